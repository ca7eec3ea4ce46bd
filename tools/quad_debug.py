"""debug aid: the quad kernel (csrc/kernel_quad.hpp) against the two-document kernel on a slice of configs[3], every draw-tier mode:
full integer state after each of two sweeps.  python tools/quad_debug.py [documents]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 31250
ref, ok = None, True
for quad, margin in ((False, 0), (True, 0), (True, -2), (True, 6), (True, -1)):
    os.environ["LLDA_QUAD"] = "on" if quad else "off"
    s, info = bench.build_sampler("synth2", dev, 0, 1, False, docs_total=docs)
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
