"""A/B of two builds of the library on one box: kernel time of the sweep and the state digests after the same number of sweeps
(they must agree: a faster kernel draws the same topics).  python tools/ab_lib.py <workload> [docs]    (LLDA_GIBBS_LIB selects
the build; tools/ab_lib.sh runs the pair).  Extra workloads: k1024, k256, k768 (dense masks)."""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    bench.WORKLOADS["k1024"] = (125000, 300, 100000, 1024, 1.0, 15625, "K = 1024")
    bench.WORKLOADS["k256"] = (250000, 300, 100000, 256, 1.0, 15625, "K = 256")
    bench.WORKLOADS["k768"] = (125000, 300, 100000, 768, 1.0, 15625, "K = 768")
    name = sys.argv[1]
    docs = int(sys.argv[2]) if len(sys.argv) > 2 else None
    dev = torch.device("cuda", 0)
    kw = {"docs_total": docs} if docs else {}
    s, info = bench.build_sampler(name, dev, 0, 1, False, **kw)
    dt, kms = bench.time_sweeps(s, 20, 3)
    st = s.status.cpu().numpy()
    print("%-16s %-22s kernel %.4f ms  %.0f M sites/s  digests %s  unsure %d exact %d" %
          (name, os.environ.get("LLDA_GIBBS_LIB", "production").split("/")[-1], kms, s.S / kms / 1e3, bench.state_checksums(s),
           int(st[1]), int(st[2])), flush=True)


if __name__ == "__main__":
    main()
