"""The narrow image's own column order (GibbsSampler(image_order=...), llda_pack_image_cols) on / off: time of a sweep (pack + kernel +
fold) and the equality of the states.  python tools/abl_image_order.py [workload ...]"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    from lda_thesis_amd.sampler import GibbsSampler
    dev = torch.device("cuda", 0)
    real_init = GibbsSampler.__init__
    for name in (sys.argv[1:] or ["synth2_sparse_scr", "synth2_sparse_hier", "synth2_sparse", "synth_wide_sparse", "abstracts"]):
        res = {}
        for on in (False, True):
            def init(self, *a, **k):
                k["image_order"] = True if on else False
                if name == "abstracts":
                    k["image"] = 8                      # (the automatic rule takes no image for a 31 MB n_kw)
                real_init(self, *a, **k)
            GibbsSampler.__init__ = init
            s, info = bench.build_sampler(name, dev, 0, 1, False)
            GibbsSampler.__init__ = real_init
            assert s.n_kw_img is not None and (s._img_src is not None) == on
            steps = 200 if name == "abstracts" else 40
            dt, kms = bench.time_sweeps(s, steps, 3)
            res[on] = (dt / steps * 1e3, kms, bench.state_checksums(s), s.z.clone(), s.image_lines_per_site)
            del s, info
            torch.cuda.empty_cache()
        same = res[False][2] == res[True][2] and bool(torch.equal(res[False][3], res[True][3]))
        print("%s | column = position: step %.3f ms (kernels %.3f) | own order: step %.3f ms (kernels %.3f) | x%.3f | same state: %s | lines per "
              "site before / after %s" % (name, res[False][0], res[False][1], res[True][0], res[True][1], res[False][0] / res[True][0], same,
                                          res[True][4]), flush=True)
        assert same


if __name__ == "__main__":
    main()
