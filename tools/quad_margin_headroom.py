"""How much of its data-dependent tier-0 margin does the quad kernel (csrc/kernel_quad.hpp) need?  tests/neartie.py plants a threshold
within 2^-28 of a prefix-sum boundary in every document (quad lane 0 / last lane / chain seam / last but one position); tier 0 calls a
planted site undecidable exactly when its fp32 error there is below the margin.  The margin is scaled down (debug_margin -10 ... -18:
1 / 1.05 = the derived bound itself, 1/2, 1/4 ... 1/256) and the planted sites tier 0 still hands over are counted: the first scale at
which one is missed brackets the largest error / margin ratio on this adversarial set.  (A missed site is decided by tier 0 ITSELF,
rightly or wrongly: the state is compared with the C oracle as well.)
python tools/quad_margin_headroom.py [K:documents[:wide_every] ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import c_oracle
from test_gpu_parity import quad_neartie_state
from lda_thesis_amd.sampler import GibbsSampler
c_oracle.build(); c_oracle.lib()
SCALES = {0: "1", -10: "1/1.05", -11: "1/2", -12: "1/4", -13: "1/8", -14: "1/16", -15: "1/32", -16: "1/64", -17: "1/128", -18: "1/256"}
print("| K | documents | int32 rows | planted | " + " | ".join("unsure at %s (wrong sites)" % v for v in SCALES.values()) + " |")
print("|---|---|---|---|" + "---|" * len(SCALES))
for spec in (sys.argv[1:] or ["512:500", "512:300:5", "256:500", "128:700", "100:600", "400:400"]):
    parts = [int(t) for t in spec.split(":")]
    K, D, wide_every = parts[0], parts[1], (parts[2] if len(parts) > 2 else 0)
    st = quad_neartie_state(K, D, wide_every)
    cs = c_oracle.CState(st["doc_off"], st["word"], st["freq"], st["z"], st["labs"], st["n_d_k"], st["n_k_v"], st["n_zk"], st["V"],
                         st["alpha"], st["beta"])
    cs.sweep(1, 4242, 0, doc_base=7, threads=4)
    cells = []
    for m in SCALES:
        s = GibbsSampler(st["doc_off"], st["word"], st["freq"], st["z"], K, st["V"], st["alpha"], st["beta"],
                         counts=dict(n_d_k=st["n_d_k"], n_k_v=st["n_k_v"], n_zk=st["n_zk"]), seed=4242, doc_base=7, commit_log=True,
                         rows16=True, quad=True)
        assert s.quad
        s.debug_margin = m
        s.sweep()
        cells.append("%d (%d)" % (int(s.status[1]), int((s.z_topics() != cs.z).sum())))
    print("| %d | %d | %d | %d | %s |" % (K, D, int(st["wide_docs"].sum()), st["n_tuned"], " | ".join(cells)), flush=True)
