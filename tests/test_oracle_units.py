"""Unit pins of the oracle's building blocks: numpy's pairwise sum, Philox4x32-10 known answers, the
keyed uniform, the keyed draw -- numpy restatement == C restatement == numpy itself."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import llda_oracle as orc


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 15, 16, 20, 100, 127, 128, 129, 136, 200, 255, 256, 392, 512, 777, 1024])
def test_pairwise_sum_is_np_sum(c_oracle, n):
    rng = np.random.default_rng(n)
    lay = orc.layout(n) if n <= 968 or n in (1024,) else None
    for _ in range(60):
        a = rng.random(n) * rng.choice([1e-6, 1e-3, 1.0, 1e3], n)
        a[rng.random(n) < 0.3] = 0.0
        s = np.sum(a)
        assert orc.pairwise_sum(a) == s
        assert c_oracle.pairwise_sum(a) == s
        if lay is not None:
            assert orc.group_sum(lay, a) == s        # the lane-local evaluation order of the kernel


def test_philox_known_answers(c_oracle):
    # Random123 kat_vectors: philox4x32-10
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = tuple(int(x) for x in orc.philox4x32_10(*ctr, *key))
        assert got == want
        assert tuple(int(x) for x in c_oracle.philox(ctr, key)) == want


@settings(max_examples=200, deadline=None)
@given(st.integers(0, 2 ** 64 - 1), st.integers(0, 2 ** 32 - 1), st.integers(0, 2 ** 32 - 1),
       st.integers(0, 2 ** 32 - 1), st.integers(0, 2 ** 32 - 1))
def test_keyed_uniform_c_equals_numpy(seed, sweep, stream, doc, site):
    import c_oracle
    u = float(orc.keyed_uniform(seed, sweep, stream, doc, site))
    assert 0.0 <= u < 1.0
    assert c_oracle.uniform(seed, sweep, stream, doc, site) == u


def test_two_sites_share_one_philox_block():
    r = orc.philox4x32_10(3, 7, 1, 9, 42, 0)
    u_even = float(orc.keyed_uniform(42, 9, 1, 7, 6))
    u_odd = float(orc.keyed_uniform(42, 9, 1, 7, 7))
    assert u_even == ((int(r[0]) >> 5) * 2.0 ** 26 + (int(r[1]) >> 6)) / 2.0 ** 53
    assert u_odd == ((int(r[2]) >> 5) * 2.0 ** 26 + (int(r[3]) >> 6)) / 2.0 ** 53


@pytest.mark.parametrize("K", [1, 3, 8, 12, 20, 40, 128, 130, 200, 392, 512, 777, 1024])
def test_draw_c_equals_numpy_and_is_a_valid_categorical(c_oracle, K):
    rng = np.random.default_rng(K)
    for _ in range(40):
        p = rng.random(K)
        p[rng.random(K) < 0.5] = 0.0
        if p.sum() == 0:
            p[rng.integers(K)] = 1.0
        p /= np.sum(p)
        for u in (0.0, rng.random(), rng.random(), 1.0 - 2.0 ** -53):
            z = orc.draw_keyed(p, u)
            assert p[z] > 0
            assert c_oracle.draw(p, u) == z


def test_draw_frequencies_follow_the_probabilities():
    K = 12
    p = np.array([0.3, 0, 0.1, 0.05, 0, 0.2, 0.05, 0, 0.1, 0.1, 0.05, 0.05])
    p = p / np.sum(p)
    us = (np.arange(20000) + 0.5) / 20000
    counts = np.bincount([orc.draw_keyed(p, u) for u in us], minlength=K)
    np.testing.assert_allclose(counts / 20000.0, p, atol=2e-4)


@pytest.mark.parametrize("K", [12, 130, 512])
def test_word_major_c_sweep_equals_the_reference_layout_sweep(c_oracle, K):
    """bench.py's strong CPU leg (llda_oracle_sweep_wm: int32, word-major -- the GPU kernels' layout) leaves the assignments and counts
    of the reference-layout C oracle (LabeledLDA.py:108-125 restated) under O3, dense label mask, several threads"""
    rng = np.random.default_rng(K)
    D, V = 40, 60
    lens = rng.integers(1, 30, size=D)
    doc_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    S = int(doc_off[-1])
    word = np.concatenate([rng.choice(V, size=n, replace=False) for n in lens]).astype(np.int32)
    freq = rng.integers(1, 5, size=S).astype(np.int32)
    z = rng.integers(0, K, size=S).astype(np.int32)
    n_d_k = np.zeros((D, K), dtype=np.int64)
    n_k_v = np.zeros((K, V), dtype=np.int64)
    doc = np.repeat(np.arange(D), lens)
    np.add.at(n_d_k, (doc, z), freq)
    np.add.at(n_k_v, (z, word), freq)
    n_zk = n_k_v.sum(axis=1)
    a = c_oracle.CState(doc_off, word, freq, z, np.ones((D, K), dtype=np.uint8), n_d_k, n_k_v, n_zk, V, 0.1, 0.01)
    b = c_oracle.WMState(doc_off, word, freq, z, n_d_k, n_k_v, n_zk, V, 0.1, 0.01)
    for i in range(3):
        a.sweep(1, 11, i, stream=2, doc_base=5, threads=2)
        b.sweep(11, i, stream=2, doc_base=5, threads=3)
        np.testing.assert_array_equal(a.z, b.z)
        np.testing.assert_array_equal(a.n_d_k, b.n_d_k)
        np.testing.assert_array_equal(a.n_k_v.T, b.n_kw)
        np.testing.assert_array_equal(a.n_zk, b.n_k)
