"""BASELINE.json configs[4]: CascadeLDA on the abstracts corpus -- wall time of go_down_tree(it=4, s=2)
(the whole ensemble of per-node Labeled-LDA sub-problems) on one MI355X.

The corpus is rebuilt from tests/golden/abstracts_d3.npz (token ids become pseudo-tokens, every depth-3
label is expanded to its prefixes as CascadeLDA.load_corpus does).  The reference's go_down_tree(4, 2) took
66.8 s on one core of the survey container; it cannot run on the GPU box.

    python tools/bench_cascade.py [--it 4] [--s 2]
"""
import argparse
import io
import json
import os
import sys
import time
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lda_thesis_amd.CascadeLDA import CascadeLDA                       # noqa: E402
from lda_thesis_amd.corpus import cascade_corpus_from_csr              # noqa: E402
from lda_thesis_amd.text import Dictionary                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--it", type=int, default=4)
    ap.add_argument("--s", type=int, default=2)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--one-by-one", action="store_true", help="train the sub-problems one after another (llda_sweep)")
    ap.add_argument("--test-it", type=int, default=0,
                    help="also time test_down_tree over the fixture's held-out documents with this many iterations")
    args = ap.parse_args()
    g = np.load(os.path.join(ROOT, "tests", "golden", "abstracts_d3.npz"))
    names = [str(x) for x in g["labelset"]]                 # index 0 is 'root'
    docs, labs, labelset = cascade_corpus_from_csr(g["doc_off"], g["word"], g["freq"], g["lab_off"], g["lab_idx"], names)
    seen = dict.fromkeys(labelset)
    dicti = Dictionary(docs)
    import torch
    from lda_thesis_amd import _native
    _native.lib()
    torch.zeros(1, device="cuda").item()                    # HIP context and library load are not part of the ensemble
    walls, first_ph = [], None
    for rep in range(args.reps):                            # rep 0 is cold (code objects, allocator), the rest warm
        np.random.seed(0)
        t0 = time.perf_counter()
        model = CascadeLDA(docs, labs, list(seen.keys()), dicti, alpha=0.1, beta=0.01, seed=1)
        t1 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            model.go_down_tree(it=args.it, s=args.s, batched=not args.one_by_one)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        walls.append(t2 - t1)
        if first_ph is None:
            first_ph = model.ph.copy()
        assert np.array_equal(first_ph, model.ph, equal_nan=True)
    tasks = model.plan_subproblems()
    lens = np.array([len(t) for t in model.doc_tups])
    sites = [int(lens[t["docs"]].sum()) for t in tasks]
    filled = int((np.nan_to_num(model.ph).sum(axis=1) > 0).sum())
    test = None
    if args.test_it:
        toff, tw, tf = g["test_doc_off"], g["test_word"], g["test_freq"]
        held = []
        for d in range(len(toff) - 1):
            toks = []
            for v, f in zip(tw[toff[d]:toff[d + 1]], tf[toff[d]:toff[d + 1]]):
                toks += ["w%05d" % v] * int(f)
            if toks:
                held.append(toks)
        model.ph = np.nan_to_num(model.ph)               # (the never-trained '' row)
        thin = max(1, args.test_it // 6)
        t3 = time.perf_counter()
        trees = model.test_down_tree_batch(held, args.test_it, thin, 0.95)
        t4 = time.perf_counter()
        some = held[:20]
        one_by_one = [model.test_down_tree(x, args.test_it, thin, 0.95) for x in some]
        t5 = time.perf_counter()
        same = all(str(a) == str(b) for a, b in zip(trees[:20], one_by_one))
        test = {"held_out_documents": len(held), "iterations": args.test_it, "batch_s": t4 - t3,
                "one_by_one_s_per_document": (t5 - t4) / len(some), "identical_trees": same}
    print(json.dumps({"metric": "CascadeLDA go_down_tree wall time", "value": min(walls[1:] or walls), "unit": "s", "n_gpus": 1,
                      "cold_first_call_s": walls[0], "warm_calls_s": walls[1:],
                      "mode": "one sub-problem at a time (llda_sweep)" if args.one_by_one else
                              "all sub-problems batched (llda_sweep_batch)",
                      "config": {"workload": "CascadeLDA on abstracts_data.csv fixture, it=%d, s=%d" % (args.it, args.s),
                                 "sub_problems": len(tasks), "sites_per_ensemble_sweep": int(sum(sites)),
                                 "largest_sub_problem_sites": int(max(sites)), "K": model.K, "V": model.V, "D": model.D},
                      "model_build_s": t1 - t0, "ph_rows_filled": filled, "test_down_tree": test,
                      "reference_cpu_s": 66.8, "reference_cpu_note": "reference go_down_tree(4, 2), 1 core, survey container "
                      "(SURVEY.md section 6); not re-measured on this host"}))


if __name__ == "__main__":
    main()
