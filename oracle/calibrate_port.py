"""Cross-calibration of bench.py's CPU baseline  --  TEST INFRASTRUCTURE, runs ONLY in the build container.

bench.py times ``llda_oracle.sweep_sequential`` (the numpy per-site loop restating LabeledLDA.py:108-125) on the
GPU box because the reference cannot travel.  This script times that port AND the unmodified reference's own
``LabeledLDA.training_iteration`` (imported from /root/reference, gensim stubbed by refshim) on the same state, one
after the other on the same core, and writes both rates to profiles/port_calibration.json.  SURVEY.md section 8(d)
asks for the two to agree within 10 %.

Usage:  python oracle/calibrate_port.py
"""
import copy
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import llda_oracle as orc          # noqa: E402
import refshim                     # noqa: E402
from lda_thesis_amd.text import Dictionary  # noqa: E402

REF_L, _ = refshim.import_reference()


def reference_model(D, N, K, V, dense, rng):
    """a reference LabeledLDA instance on a synthetic corpus of D documents x N distinct words"""
    vocab = ["w%05d" % i for i in range(V)]
    p = 1.0 / np.arange(1, V + 1)
    p /= p.sum()
    docs = [[vocab[i] for i in sorted(rng.choice(V, size=N, replace=False, p=p))] for _ in range(D)]
    names = ["L%03d" % i for i in range(K - 1)]
    if dense:
        labs = [list(names) for _ in range(D)]
    else:
        labs = [list(rng.choice(names, size=rng.integers(1, 8), replace=False)) for _ in range(D)]
    dicti = Dictionary(docs)
    np.random.seed(3)
    return REF_L.LabeledLDA(docs, labs, list(names), dicti, 0.1, 0.01)


def port_state(m):
    st = orc.State([list(d) for d in m.docs], [list(f) for f in m.freqs], m.labs.copy(), m.V, m.alpha, m.beta,
                   [np.asarray(z).copy() for z in m.z_dn])
    st.n_k_v, st.n_zk, st.n_d_k = m.n_k_v.copy(), m.n_zk.copy(), m.n_d_k.copy()
    return st


def main():
    rng = np.random.default_rng(11)
    out = {"host": os.uname().nodename, "cores_used": 1, "cases": []}
    for name, D, N, K, V, dense in (("K=512 dense (synth2-like)", 300, 300, 512, 20000, True),
                                    ("K=128 dense (synth1-like)", 600, 200, 128, 20000, True),
                                    ("K=392, 1-7 labels per document (abstracts-like)", 2000, 48, 392, 10000, False)):
        m = reference_model(D, N, K, V, dense, rng)
        sites = sum(len(d) for d in m.docs)
        st = port_state(m)
        ref = copy.deepcopy(m)
        np.random.seed(1)
        t0 = time.perf_counter()
        ref.training_iteration()
        t_ref = time.perf_counter() - t0
        np.random.seed(1)
        t0 = time.perf_counter()
        orc.sweep_sequential(st, None)
        t_port = time.perf_counter() - t0
        same = bool(np.array_equal(st.n_k_v, ref.n_k_v) and np.array_equal(st.n_d_k, ref.n_d_k))
        out["cases"].append({"case": name, "sites": sites, "reference_Msites_s": sites / t_ref / 1e6,
                             "port_Msites_s": sites / t_port / 1e6, "port_over_reference": t_ref / t_port,
                             "identical_counts_after_the_sweep": same})
        print(out["cases"][-1], flush=True)
    with open(os.path.join(ROOT, "profiles", "port_calibration.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
