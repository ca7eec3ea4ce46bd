"""Kernel time of the dense sweep when the n_kw rows fit in L2: the same corpus shape with a small vocabulary
(valid data, valid chain: what the kernel would run at if no row came from HBM).  python tools/abl_vocab.py K N V docs"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    K, N, V, docs = (int(a) for a in sys.argv[1:5])
    bench.WORKLOADS["abl"] = (docs, N, V, K, 1.0, 1000, "ablation")
    dev = torch.device("cuda", 0)
    s, info = bench.build_sampler("abl", dev, 0, 1, False)
    for _ in range(3):
        s.sweep()
    s.kernel_events = []
    for _ in range(20):
        s.sweep()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in s.kernel_events]
    print("K %d N %d V %d docs %d: kernel ms %.4f Msites/s %.0f  tiers %s" % (K, N, V, docs, sum(ms) / len(ms), s.S / (sum(ms) / len(ms)) / 1e3,
          s.tier_counts() if hasattr(s, "tier_counts") else ""))


if __name__ == "__main__":
    main()
