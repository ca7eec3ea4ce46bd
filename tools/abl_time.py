"""kernel time of the dense sweep for an ablation build of the library (LLDA_GIBBS_LIB=...): no result checks."""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    name, docs = sys.argv[1], int(sys.argv[2])
    dev = torch.device("cuda", 0)
    s, info = bench.build_sampler(name, dev, 0, 1, False, docs_total=docs)
    for _ in range(3):
        s.sweep()
    s.kernel_events = []
    for _ in range(30):
        s.sweep()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in s.kernel_events]
    print(name, os.environ.get("LLDA_GIBBS_LIB", "production").split("/")[-1], "kernel ms %.4f" % (sum(ms) / len(ms)),
          "Msites/s %.0f" % (s.S / (sum(ms) / len(ms)) / 1e3))


if __name__ == "__main__":
    main()
