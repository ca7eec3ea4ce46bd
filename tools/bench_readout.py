"""Cost of the thinning read-outs of run_training (reference LabeledLDA.py:131-153) on the abstracts corpus
(D=4171, K=392): phi, theta, running means and guards evaluated on the device (llda_readout_phi / _theta)
against the same expressions evaluated by numpy on counts copied to the host (what the drop-in did before).

    python tools/bench_readout.py [--iters 200] [--thinning 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lda_thesis_amd.sampler import GibbsSampler      # noqa: E402


def make(g):
    return GibbsSampler(g["doc_off"], g["word"].astype(np.int32), g["freq"].astype(np.int32),
                        g["z_init"].astype(np.int64), int(g["K"]), int(g["V"]), float(g["alpha"]),
                        float(g["beta"]), labs=(g["lab_off"], g["lab_idx"].astype(np.int64)), counts=None,
                        seed=int(g["seed"]))


def labs_matrix(g):
    labs = np.zeros((int(g["D"]), int(g["K"])))
    rows = np.repeat(np.arange(int(g["D"])), np.diff(g["lab_off"]))
    labs[rows, g["lab_idx"]] = 1.0
    return labs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--thinning", type=int, default=3)
    a = ap.parse_args()
    g = np.load(os.path.join(ROOT, "tests", "golden", "abstracts_d3.npz"))
    labs = labs_matrix(g)
    res = {}
    for where in ("device", "host"):
        s = make(g)
        s.sweep()
        torch.cuda.synchronize()
        ph = th = None
        t0 = time.perf_counter()
        for n in range(a.iters):
            s.sweep()
            if (n + 1) % a.thinning:
                continue
            k = (n + 1) / a.thinning
            keep, share = ((k - 1) / k, 1 / k) if k > 1 else (None, None)
            if where == "device":
                flags = torch.zeros((1,), dtype=torch.int32, device=s.device)
                ph = s.phi(ph if k > 1 else None, keep, share, flags)
                th = s.theta(th if k > 1 else None, keep, share)
                assert int(flags.item()) == 0
            else:
                cur_ph = (s.n_k_v() + s.beta) / (s.n_zk()[:, np.newaxis] + s.V * s.beta)
                num = s.n_d_k() + labs * s.alpha
                cur_th = num / num.sum(axis=1)[:, np.newaxis]
                ph = cur_ph if k == 1 else keep * ph + (share * cur_ph)
                th = cur_th if k == 1 else keep * th + (share * cur_th)
                assert not (ph < 0).any() and not np.isnan(ph).any() and not (ph.sum(axis=0) == 0).any()
        torch.cuda.synchronize()
        res[where] = time.perf_counter() - t0
        res[where + "_ph"] = ph.cpu().numpy() if where == "device" else ph
        res[where + "_th"] = th.cpu().numpy() if where == "device" else th
    same = bool(np.array_equal(res["device_ph"], res["host_ph"]) and np.array_equal(res["device_th"], res["host_th"]))
    print(json.dumps({"workload": "abstracts d=3: %d sweeps, read-out every %d" % (a.iters, a.thinning),
                      "seconds_readouts_on_device": round(res["device"], 3),
                      "seconds_readouts_on_host": round(res["host"], 3), "identical_means": same}))


if __name__ == "__main__":
    main()
