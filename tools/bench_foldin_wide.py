"""Test-time fold-in (llda_foldin) on a narrow and on a wide layout: 400 held-out documents x 50 sites x 100 sweeps.
Measured on MI355X: K = 512 0.016 s (125 M site-draws/s), K = 2048 0.12 s (17 M).  python tools/bench_foldin_wide.py"""


def main():
    import os, sys, time, numpy as np, torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lda_thesis_amd.foldin import fold_in
    rng = np.random.default_rng(1)
    for K in (512, 2048):
        V = 3000
        ph = rng.random((K, V)) ** 4
        ph /= ph.sum(axis=1, keepdims=True)
        tups = []
        for d in range(400):
            ids = np.sort(rng.choice(V, size=50, replace=False)).tolist()
            tups.append(list(zip(ids, [1] * 50)))
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = fold_in(ph, 0.1, tups, 100, 10, 5)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        print("K", K, "400 docs x 50 sites x 100 sweeps: %.3f s  (%.2f M site-draws/s)" % (dt, 400 * 50 * 100 / dt / 1e6))


if __name__ == "__main__":
    main()
