"""A/B of the sparse-label kernel with and without its narrow image (llda_sweep_args.n_kw_img): kernel time (HIP events around
llda_pack_image + llda_sweep), throughput and the state digests after the same number of sweeps (they must agree).
    python tools/abl_image.py [workload ...]       default: synth2_sparse synth_wide_sparse synth2_sparse_hier"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    docs = 0
    if "--docs" in sys.argv:                       # e.g. --docs 1000000: the sparse variant of configs[3] at its full size
        i = sys.argv.index("--docs")
        docs = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    names = sys.argv[1:] or ["synth2_sparse", "synth_wide_sparse", "synth2_sparse_hier"]
    dev = torch.device("cuda", 0)
    for name in names:
        for image in ("0", "8", "16", "auto"):
            if image == "auto":
                os.environ.pop("LLDA_IMAGE", None)
            else:
                os.environ["LLDA_IMAGE"] = image
            s, info = bench.build_sampler(name, dev, 0, 1, False, docs_total=docs)
            r8, r16 = s._image_escape_rates()
            dt, kms = bench.time_sweeps(s, 50 if not docs else 20, 5)
            bits = 0 if s.n_kw_img is None else 8 * s.n_kw_img.element_size()
            print("%-20s image %-4s -> %2d bits  kernel %.4f ms  %.0f M sites/s (wall %.0f)  escapes8 %.3f escapes16 %.4f  digests %s" %
                  (name, image, bits, kms, s.S / kms / 1e3, s.S * (50 if not docs else 20) / dt / 1e6, r8, r16, bench.state_checksums(s)), flush=True)
            del s, info
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
