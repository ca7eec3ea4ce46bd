"""Where a wavefront of the quad kernel (csrc/kernel_quad.hpp) spends its time: a -DQUAD_PROFILE build stamps the shader clock at the
phase boundaries of every site in wavefront 0 of workgroup 0 and adds the differences up in status[8 + phase].
    hipcc ... -DQUAD_PROFILE -o tools/bin/libllda_qprof.so;  LLDA_GIBBS_LIB=$PWD/tools/bin/libllda_qprof.so python tools/quad_phase_profile.py [documents [workload]]"""


def main():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import bench
    dev = torch.device("cuda", 0)
    s, info = bench.build_sampler(sys.argv[2] if len(sys.argv) > 2 else "synth2", dev, 0, 1, False,
                                  docs_total=int(sys.argv[1]) if len(sys.argv) > 1 else 125000)
    assert s.quad
    s.status = torch.zeros((64,), dtype=torch.int32, device=dev)
    for _ in range(3):
        s.sweep()
    torch.cuda.synchronize()
    s.status.zero_()
    n = 5
    for _ in range(n):
        s.sweep()
    torch.cuda.synchronize()
    st = s.status.cpu().numpy()
    sites = int(st[16])
    names = ["front (factor reads issued, uniform)", "chains + scan + search + pick", "cold tiers",
             "decode + the update's LDS reads issued", "records / z of the sites ahead issued", "own count out + next row -> fp32 (waits for the row)",
             "row prefetch issued, update written, commit", "-"]
    tot = float(st[8:16].sum())
    print("wave iterations stamped: %d; clock ticks per iteration: %.1f (s_memtime ticks)" % (sites, tot / max(sites, 1)))
    for k in range(7):
        print("  phase %d %-70s %8.2f ticks per iteration  %5.1f %%" % (k, names[k], st[8 + k] / max(sites, 1), 100.0 * st[8 + k] / max(tot, 1)))


if __name__ == "__main__":
    main()
