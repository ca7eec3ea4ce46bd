"""16-bit rows of n_kw (llda_sweep_args.n_kw16) on / off: kernel time of the sweep (the 16-bit image
is refreshed inside the timed bracket) and the equality of the states, for the dense K = 512 / 1024 workloads.
python tools/abl_rows16.py [workload ...]"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    bench.WORKLOADS["k1024"] = (125000, 300, 100000, 1024, 1.0, 15625, "K = 1024")
    bench.WORKLOADS["zipf_v300k"] = (125000, 300, 300000, 512, 1.0, 15625, "Zipf words, V = 300k: n_kw 614 MB")
    bench.WORKLOADS["zipf_v1m"] = (125000, 300, 1000000, 512, 1.0, 15625, "Zipf words, V = 1M: n_kw 2 GB")
    bench.WORKLOADS["zipf_v20k"] = (125000, 300, 20000, 512, 1.0, 15625, "Zipf words, V = 20k: n_kw 41 MB")
    bench.WORKLOADS["k1024_v20k"] = (62500, 300, 20000, 1024, 1.0, 15625, "K = 1024, Zipf words, V = 20k: n_kw 82 MB")
    bench.WORKLOADS["k1024_v500k"] = (125000, 300, 500000, 1024, 1.0, 15625, "K = 1024, Zipf words, V = 500k: n_kw 2 GB")
    dev = torch.device("cuda", 0)
    for name in (sys.argv[1:] or ["synth2", "synth2_hostile", "k1024"]):
        res = {}
        for on in (False, True):
            s, info = bench.build_sampler(name, dev, 0, 1, False, rows16=on)
            assert (s.n_kw16 is not None) == on
            dt, kms = bench.time_sweeps(s, 10, 2)
            st = s.status.cpu().numpy()
            res[on] = (kms, s.S / kms / 1e3, bench.state_checksums(s), s.z.clone(), int(st[1]), int(st[2]),
                       float(s.row16.float().mean()) if on else 0.0,
                       float(s.row16[s.word.long()].float().mean()) if on else 0.0)
            del s, info
            torch.cuda.empty_cache()
        same = res[False][2] == res[True][2] and bool(torch.equal(res[False][3], res[True][3]))
        print("%s | int32 rows: %.3f ms %.0f M sites/s | 16-bit rows: %.3f ms %.0f M sites/s (%.1f %% of the words, %.1f %% of the sites) | "
              "x%.3f | same state after 12 sweeps: %s | unsure %d / %d" %
              (name, res[False][0], res[False][1], res[True][0], res[True][1], 100 * res[True][6], 100 * res[True][7],
               res[False][0] / res[True][0], same, res[False][4], res[True][4]), flush=True)
        assert same


if __name__ == "__main__":
    main()
