"""debug aid: the quad kernel (csrc/kernel_quad.hpp) against the kernel the sampler takes without it (K = 512: two documents per
wavefront on 16-bit rows; K = 128, 256: the general kernel on int32 rows) on a slice of a bench workload, every draw-tier mode: full
integer state after each of two sweeps.  python tools/quad_debug.py [documents [workload]]     (synth2 | synth1 | synth_k256)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 31250
wl = sys.argv[2] if len(sys.argv) > 2 else "synth2"
ref, ok = None, True
for quad, margin in ((False, 0), (True, 0), (True, -2), (True, 6), (True, -1)):
    os.environ["LLDA_QUAD"] = "on" if quad else "off"
    s, info = bench.build_sampler(wl, dev, 0, 1, False, docs_total=docs)
    assert s.quad == quad, (s.quad, quad)
    s.debug_margin = margin
    states = []
    for i in range(2):
        s.sweep()
        states.append((s.z.clone(), s.n_dk.clone(), s.n_kw.clone(), s.n_k.clone()))
    torch.cuda.synchronize()
    if ref is None:
        ref = states
    d = [[int((a != b).sum()) for a, b in zip(r, c)] for r, c in zip(ref, states)]
    ok = ok and not any(any(x) for x in d)
    print("quad", quad, "margin", margin, "status", s.status.cpu().numpy().tolist(), "differences (z, n_dk, n_kw, n_k) after sweep 1:", d[0], "after 2:", d[1], flush=True)
    del s
print("ALL EQUAL" if ok else "MISMATCH")
