"""Sparse-label sweep kernel with a small vocabulary (rows resident in L2) vs the real one: how much of its time is memory?
python tools/abl_vocab_sparse.py V docs"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    V, docs = int(sys.argv[1]), int(sys.argv[2])
    w = list(bench.WORKLOADS["synth2_sparse"])
    w[0], w[2] = docs, V
    bench.WORKLOADS["synth2_sparse"] = tuple(w)
    dev = torch.device("cuda", 0)
    s, info = bench.build_sampler("synth2_sparse", dev, 0, 1, False)
    for _ in range(3):
        s.sweep()
    s.kernel_events = []
    for _ in range(20):
        s.sweep()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in s.kernel_events]
    print("sparse V %d docs %d: kernel ms %.4f Msites/s %.0f" % (V, docs, sum(ms) / len(ms), s.S / (sum(ms) / len(ms)) / 1e3))


if __name__ == "__main__":
    main()
