"""Four documents per wavefront (llda_sweep_args.row16, csrc/kernel_quad.hpp) against the two-document 16-bit-row kernel: kernel
time of the sweep (the image is refreshed inside the timed bracket) and the equality of the states.
python tools/abl_quad.py [workload[:documents] ...]"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from lda_thesis_amd.sampler import GibbsSampler
    dev = torch.device("cuda", 0)
    for spec in (sys.argv[1:] or ["synth2:125000", "synth2"]):
        name, _, docs = spec.partition(":")
        res = {}
        for quad in (False, True):
            os.environ["LLDA_QUAD"] = "on" if quad else "off"
            s, info = bench.build_sampler(name, dev, 0, 1, False, docs_total=int(docs or 0))
            assert s.quad == quad, (s.quad, quad)
            dt, kms = bench.time_sweeps(s, 10, 2)
            st = s.status.cpu().numpy()
            res[quad] = (kms, s.S / kms / 1e3, bench.state_checksums(s), s.z.clone(), int(st[1]), int(st[2]), dt / 10 * 1e3)
            del s, info
            torch.cuda.empty_cache()
        same = res[False][2] == res[True][2] and bool(torch.equal(res[False][3], res[True][3]))
        print("%s | two documents per wavefront: %.3f ms %.0f M sites/s (step %.3f ms) | four: %.3f ms %.0f M sites/s (step %.3f ms) | "
              "x%.3f | same state after 12 sweeps: %s | unsure %d / %d, exact %d / %d" %
              (spec, res[False][0], res[False][1], res[False][6], res[True][0], res[True][1], res[True][6],
               res[False][0] / res[True][0], same, res[False][4], res[True][4], res[False][5], res[True][5]), flush=True)
        assert same
    os.environ.pop("LLDA_QUAD", None)


if __name__ == "__main__":
    main()
