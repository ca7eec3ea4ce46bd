"""Thinning read-outs on the device (llda_readout_phi / llda_readout_theta through the C ABI): phi, theta,
SubLDA's get_ph, their running means and the three guards of reference LabeledLDA.py:146-153 -- bit for
bit against the reference's outputs (golden vectors) and against numpy's own evaluation of the reference's
expressions on the same counts."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

TINY = golden_names("tiny_")


def final_state_sampler(g, **kw):
    """sampler holding the counts the reference had after the golden's last sweep."""
    from lda_thesis_amd.sampler import GibbsSampler
    key = "o3_s%d_" % int(g["sweeps"])
    c = dict(n_d_k=g[key + "n_d_k"], n_k_v=g[key + "n_k_v"], n_zk=g[key + "n_zk"])
    return GibbsSampler(g["doc_off"], g["word"], g["freq"], g[key + "z"], int(g["K"]), int(g["V"]),
                        float(g["alpha"]), float(g["beta"]), labs=g["labs"], counts=c, seed=int(g["seed"]), **kw)


@pytest.mark.parametrize("name", TINY)
def test_phi_theta_match_reference(name):
    g = load_golden(name)
    s = final_state_sampler(g)
    np.testing.assert_array_equal(s.phi().cpu().numpy(), g["o3_phi"])
    np.testing.assert_array_equal(s.theta().cpu().numpy(), g["o3_theta"])


@pytest.mark.parametrize("name", TINY)
def test_running_means_round_like_numpy(name):
    """out = keep*out + (share*cur): two products and one sum, each rounded (LabeledLDA.py:144-145)."""
    g = load_golden(name)
    s = final_state_sampler(g)
    rng = np.random.default_rng(5)
    for cur, fn in ((g["o3_phi"], s.phi), (g["o3_theta"], s.theta)):
        old = rng.random(cur.shape)
        for n in (2.0, 3.0, 7.0):
            keep, share = (n - 1) / n, 1 / n
            dev = torch.from_numpy(old.copy()).cuda()
            fn(dev, keep, share)
            np.testing.assert_array_equal(dev.cpu().numpy(), keep * old + (share * cur))


@pytest.mark.parametrize("name", ["tiny_k05", "tiny_k130", "tiny_k512", "sublda"])
def test_get_ph_rows(name):
    """SubLDA.get_ph: n_k_v / n_k_v.sum(axis=1) (CascadeLDA.py:394-395), empty rows give NaN as numpy does."""
    g = load_golden(name)
    s = final_state_sampler(g)
    n_k_v = s.n_k_v()
    with np.errstate(divide="ignore", invalid="ignore"):
        want = n_k_v / n_k_v.sum(axis=1, keepdims=True)
    np.testing.assert_array_equal(s.ph_rows().cpu().numpy(), want)


def test_guards():
    from lda_thesis_amd import _native
    g = load_golden("tiny_k40")
    s = final_state_sampler(g)
    K, V = s.K, s.V

    def flags_after(out, keep=None, share=None):
        f = torch.zeros((1,), dtype=torch.int32, device="cuda")
        s.phi(out, keep, share, f)
        return int(f.item())

    out = torch.empty((K, V), dtype=torch.float64, device="cuda")
    assert flags_after(out) == 0                                            # a healthy phi
    neg = torch.full((K, V), -5.0, dtype=torch.float64, device="cuda")
    assert flags_after(neg, 0.5, 0.5) == _native.READOUT_NEGATIVE
    nan = torch.zeros((K, V), dtype=torch.float64, device="cuda")
    nan[3, 7] = float("nan")
    assert flags_after(nan, 0.5, 0.5) == _native.READOUT_NAN
    # a word whose column is all zero: only reachable through the mean's coefficients
    assert flags_after(torch.zeros((K, V), dtype=torch.float64, device="cuda"), 0.0, 0.0) == _native.READOUT_NO_LOAD


def test_wide_vocabulary_tiles():
    """V not a multiple of the 64-word tile, KP not a multiple of 64: every (k, v) written exactly once."""
    from lda_thesis_amd.corpus import synthetic_corpus
    from lda_thesis_amd.sampler import GibbsSampler
    for K, V in ((12, 1000), (96, 333), (200, 65), (1031, 70), (5000, 65), (7688, 40)):      # the last three: wide layouts
        doc_off, word, freq, z = synthetic_corpus(50, 20, V, K, K, "cuda")
        s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, seed=1)
        out = torch.full((K, V), -1.0, dtype=torch.float64, device="cuda")
        s.phi(out)
        want = (s.n_k_v() + 0.01) / (s.n_zk()[:, np.newaxis] + V * 0.01)
        np.testing.assert_array_equal(out.cpu().numpy(), want)
        num = s.n_d_k() + np.ones((50, K)) * 0.1
        np.testing.assert_array_equal(s.theta().cpu().numpy(), num / num.sum(axis=1)[:, np.newaxis])


def test_perplexity_of_wide_layouts():
    """llda_loglik on layouts with more than 8 pairwise leaves (one wavefront per document; 128 KB of LDS at K = 7688)
    against numpy's evaluation of LabeledLDA.py:256-265 on the same counts."""
    from lda_thesis_amd.corpus import synthetic_corpus
    from lda_thesis_amd.sampler import GibbsSampler
    for K, V in ((1100, 60), (7688, 40)):
        doc_off, word, freq, z = synthetic_corpus(30, 20, V, K, K, "cuda")
        s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, seed=1)
        s.sweep()
        phi = (s.n_k_v() + 0.01) / (s.n_zk()[:, np.newaxis] + V * 0.01)
        num = s.n_d_k() + 0.1
        th = num / num.sum(axis=1)[:, np.newaxis]
        off, w = doc_off.cpu().numpy(), word.cpu().numpy()
        ll = 0.0
        for d in range(30):
            for i in range(off[d], off[d + 1]):
                ll -= np.log(np.inner(phi[:, w[i]], th[d]))
        assert abs(s.perplexity() / np.exp(ll / off[-1]) - 1) < 1e-9
