"""Wide layouts (K with more than 8 pairwise leaves): kernel time of the sweep for the kernel variants llda_sweep's
debug_margin selects -- 0 production (fp32 tier 0: float factors + int16 count changes in LDS, rare tiers on a scratch
row), -7 the fp32 tier with fp32 factors only in LDS (rare tiers on the scratch row), -6 with fp64 factors in LDS, -5 the fp64 kernel with the row in registers and int16 count changes, -4 the same with LDS copies of the counts,
-3 LDS-only kernel.
python tools/abl_wide.py K [N V docs]"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    K = int(sys.argv[1])
    N, V, docs = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (100, 20000, 20000)
    bench.WORKLOADS["abl"] = (docs, N, V, K, 1.0, 1000, "ablation")
    dev = torch.device("cuda", 0)
    s, info = bench.build_sampler("abl", dev, 0, 1, False)
    out = []
    for dm in (0, -7, -6, -5, -4, -3):
        s.debug_margin = dm
        for _ in range(2):
            s.sweep()
        s.kernel_events = []
        for _ in range(8):
            s.sweep()
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in s.kernel_events]
        st = s.status.cpu().numpy()
        out.append("%d: %.2f ms %.0f M/s (unsure %d exact %d)" % (dm, sum(ms) / len(ms), s.S / (sum(ms) / len(ms)) / 1e3, int(st[1]), int(st[2])))
        s.status.zero_()
    print("K %d tiers %d T %d | " % (K, s.layout.NT, s.layout.T) + " | ".join(out))


if __name__ == "__main__":
    main()
