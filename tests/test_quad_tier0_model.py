"""The hand-derived error bound of the quad kernels' tier 0, checked on the CPU (no GPU needed).

lda_thesis_amd/csrc/kernel_quad.hpp decides 99.8 % of the sites of the bench's workload in fp32 and is allowed to only where no prefix
sum lies within  m = 1.05 v (32 L + 39 t + 37 P + 0.25 total)  of the threshold; its header derives, by hand, that every compared
difference is within  v (31.1 L + 38.2 t + 36.1 P + 0.125 total)  of its real-number value (v = 2^-24; L the lane's total, P the lanes
before it, t = u total).  oracle/quad_tier0_model.c replays tier 0 in fp32 in the kernel's association order (packed chains, row_shr
scan, rotate / xor totals, one-fma target) next to 80-bit arithmetic and reports, for EVERY position of every lane of every vector, the
error of the compared difference relative to the margin and to the bound -- on heavy-tailed, sparse, concentrated and
beyond-2^24 count vectors, thresholds both uniform and planted next to a prefix boundary, the hardware reciprocal modelled as exact and as
1 ulp off either way.  /root/reference/LabeledLDA.py:113-119 is what the decision must reproduce."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(os.path.dirname(HERE), "oracle")


@pytest.fixture(scope="module")
def model():
    so = os.path.join(ORACLE, "libquad_tier0_model.so")
    src = os.path.join(ORACLE, "quad_tier0_model.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE, "-B", "libquad_tier0_model.so"], stdout=subprocess.DEVNULL)
    L = ctypes.CDLL(so)
    P = ctypes.c_void_p
    L.quad_tier0_model.restype = ctypes.c_int
    L.quad_tier0_model.argtypes = [ctypes.c_int, ctypes.c_int64, P, P, P, P, P, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_int, ctypes.c_uint32, ctypes.c_float, ctypes.c_float, P, P, P, P]
    return L


def run_model(L, LB, x, nd, nk, valid, u, alpha, beta, vbeta, rcp_mode=0, seed=0, margin_rel=0.0, margin_data=1.0):
    n, KP = x.shape
    assert KP == 32 << LB
    x, nd, nk = (np.ascontiguousarray(a, dtype=np.int32) for a in (x, nd, nk))
    u = np.ascontiguousarray(u, dtype=np.float64)
    v = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
    pos = np.zeros(n, dtype=np.int32)
    uns = np.zeros(n, dtype=np.uint8)
    pex = np.zeros(n, dtype=np.int32)
    rat = np.zeros((n, 8), dtype=np.float64)
    rc = L.quad_tier0_model(LB, n, x.ctypes.data, nd.ctypes.data, nk.ctypes.data, None if v is None else v.ctypes.data, u.ctypes.data,
                            alpha, beta, vbeta, rcp_mode, seed, margin_rel, margin_data, pos.ctypes.data, uns.ctypes.data,
                            pex.ctypes.data, rat.ctypes.data)
    assert rc == 0
    return pos, uns.astype(bool), pex, rat


def make_vectors(rng, n, KP, kind, valid=None):
    """count vectors in draw order: x (the word's row minus the site's own count), nd (the document's counts), nk (the topic totals)."""
    if kind == "heavy":                      # tests/neartie.py's columns: 1 ... 65 000 log-uniform, 30 % zeros
        x = np.floor(np.exp(rng.uniform(0.0, np.log(65000.0), size=(n, KP))))
        x[rng.random((n, KP)) < 0.3] = 0
        nd = np.where(rng.random((n, KP)) < 0.1, rng.integers(0, 100, size=(n, KP)), rng.integers(0, 8, size=(n, KP)))
    elif kind == "sparse":                   # a real corpus: most counts 0, a few small, the document sits in a handful of topics
        x = np.where(rng.random((n, KP)) < 0.05, rng.integers(1, 40, size=(n, KP)), 0)
        nd = np.where(rng.random((n, KP)) < 0.03, rng.integers(1, 60, size=(n, KP)), 0)
    elif kind == "peaked":                   # one slot carries almost everything (the lane total L ~ total; the rest tiny)
        x = rng.integers(0, 3, size=(n, KP))
        nd = rng.integers(0, 3, size=(n, KP))
        hot = rng.integers(0, KP, size=n)
        x[np.arange(n), hot] = rng.integers(20000, 65000, size=n)
        nd[np.arange(n), hot] = rng.integers(1000, 60000, size=n)
    elif kind == "flat":                     # all slots alike: the longest possible run of equal-magnitude additions
        x = rng.integers(900, 1100, size=(n, KP))
        nd = rng.integers(90, 110, size=(n, KP))
    elif kind == "wide":                     # counts beyond 65 535 and beyond 2^24: the int32 read (+ 2 v in the bound)
        x = np.floor(np.exp(rng.uniform(0.0, np.log(3.0e7), size=(n, KP))))
        x[rng.random((n, KP)) < 0.3] = 0
        nd = np.where(rng.random((n, KP)) < 0.1, rng.integers(0, 100, size=(n, KP)), rng.integers(0, 8, size=(n, KP)))
    elif kind == "ramp":                     # counts grow along the draw order: every addition meets a sum of its own size
        base = np.exp(np.linspace(0.0, np.log(60000.0), KP))[None, :] * rng.uniform(0.5, 1.0, size=(n, KP))
        x = np.floor(base if rng.random() < 0.5 else base[:, ::-1])
        nd = rng.integers(0, 50, size=(n, KP))
    else:
        raise ValueError(kind)
    nk = np.floor(np.exp(rng.uniform(np.log(1e3), np.log(3e7), size=(n, KP))))
    if valid is not None:
        x, nd = x * valid[None, :], nd * valid[None, :]
    return x.astype(np.int64), nd.astype(np.int64), nk.astype(np.int64)


def planted_u(rng, x, nd, nk, valid, alpha, beta, vbeta, spread):
    """uniforms that put the threshold within `spread` (relative to the total, log-uniform down to 2^-34) of a prefix boundary."""
    w = (nd + alpha) * ((x + beta) / (nk + vbeta))
    if valid is not None:
        w = w * valid[None, :]
    cum = np.cumsum(w, axis=1)
    n, KP = x.shape
    k = rng.integers(0, KP - 1, size=n)
    b = cum[np.arange(n), k] / cum[:, -1]
    eps = np.exp(rng.uniform(np.log(2.0 ** -34), np.log(spread), size=n)) * rng.choice([-1.0, 1.0], size=n)
    return np.clip(b + eps, 0.0, 1.0 - 2.0 ** -53)


KINDS = ("heavy", "sparse", "peaked", "flat", "wide", "ramp")
PRIORS = ((0.1, 0.01, 1000.0), (50.0 / 512, 0.01, 0.01 * 100000), (1e-6, 1e-6, 1e-4), (0.5, 2.0, 2.0e7), (1e-3, 1e-3, 3.0))


def pad_valid(K, LB):
    """draw-order validity of a K < KP layout of the quad geometry LB (None for K = KP)."""
    import llda_oracle as orc
    lay = orc.layout(K)
    assert lay.T == 16 and lay.G == 2 << LB
    v = (lay.slot_topic >= 0).astype(np.uint8)             # index = standard lane * 16 + slot = quad lane * 32 + chain * 16 + slot
    return None if v.all() else v


@pytest.mark.parametrize("LB,K", [(4, 512), (3, 256), (2, 128), (4, 400), (2, 100)])
def test_tier0_error_stays_inside_the_derived_bound(model, LB, K):
    """|computed difference - real difference| <= bound <= margin / 1.05, term by term, for every position of every lane: K = 512 (the
    kernel bench.py times) on 1.08 million vectors (540 000 distinct count vectors x the two reciprocal models: 5.5 * 10^8 positions),
    the other geometries on 135 000 each."""
    KP = 32 << LB
    valid = pad_valid(K, LB)
    rng = np.random.default_rng(1000 + K)
    n = 48000 if K == 512 else 6000 if KP == 512 else 12000
    worst = np.zeros(8)
    worst[6] = 1e300
    sure_total = wrong = 0
    for kind in KINDS:
        for alpha, beta, vbeta in PRIORS:
            for planted in (False, True):
                x, nd, nk = make_vectors(rng, n // 4 if planted else n // 8, KP, kind, valid)
                u = planted_u(rng, x, nd, nk, valid, alpha, beta, vbeta, 2.0 ** -17) if planted else rng.random(x.shape[0])
                for rcp_mode in (0, 1):
                    pos, uns, pex, rat = run_model(model, LB, x, nd, nk, valid, u, alpha, beta, vbeta, rcp_mode, seed=K)
                    assert not rat[:, 7].any(), "%s: a chain is not sorted / the lanes disagree about the total" % kind
                    sure = ~uns
                    sure_total += int(sure.sum())
                    wrong += int((pos[sure] != pex[sure]).sum())
                    worst[:6] = np.maximum(worst[:6], rat[:, :6].max(axis=0))
                    worst[6] = min(worst[6], rat[:, 6].min())
                    if planted:
                        # a threshold within 2^-17 of a boundary is mostly inside the margin: tier 0 must notice most of them
                        assert uns.mean() > 0.2
    # the decision: whenever every lane was sure, tier 0 named the real-number draw
    assert sure_total > 1000 and wrong == 0
    # the derivation, term by term (kernel_quad.hpp): prefixes within 28.1 v L, scan and total within 33.1 v, u~ within 2^-27 + v u,
    # every compared difference within the bound, the bound within the margin
    assert worst[2] <= 1.0, "prefix error / (28.1 v L) = %.3f" % worst[2]
    assert worst[3] <= 1.0 and worst[4] <= 1.0, "scan / total error over 33.1 v: %.3f / %.3f" % (worst[3], worst[4])
    assert worst[5] <= 1.0, "u32 error = %.3f of its bound" % worst[5]
    assert worst[1] <= 1.0, "compared difference off by %.3f of the derived bound" % worst[1]
    assert worst[6] >= 1.0, "margin / bound = %.4f somewhere" % worst[6]
    assert worst[0] < 1.0 / 1.02, "compared difference off by %.3f of the margin" % worst[0]
    print("K = %d: max error / margin %.3f, / bound %.3f, prefix %.3f, scan %.3f, total %.3f, u %.3f; min margin / bound %.4f; %d sure decisions"
          % (K, worst[0], worst[1], worst[2], worst[3], worst[4], worst[5], worst[6], sure_total))


def test_model_has_teeth(model):
    """the same check fails when the margin is scaled down far enough: with 1 / 64 of the margin, planted ties are decided in fp32
    and some of them wrongly -- so a bound that were too small by such a factor would not pass the test above."""
    rng = np.random.default_rng(7)
    x, nd, nk = make_vectors(rng, 20000, 512, "heavy")
    u = planted_u(rng, x, nd, nk, None, 0.1, 0.01, 1000.0, 2.0 ** -26)
    pos, uns, pex, rat = run_model(model, 4, x, nd, nk, None, u, 0.1, 0.01, 1000.0, margin_data=1.0 / 64)
    sure = ~uns
    assert sure.sum() > 100 and (pos[sure] != pex[sure]).sum() > 0 and rat[:, 0].max() > 1.0
    pos, uns, pex, rat = run_model(model, 4, x, nd, nk, None, u, 0.1, 0.01, 1000.0)
    sure = ~uns
    assert (pos[sure] == pex[sure]).all() and rat[:, 0].max() < 1.0
