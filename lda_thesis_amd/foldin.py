"""Test-time fold-in sampler on the device: the reference's ``prep4test`` / ``run_test``
(/root/reference/LabeledLDA.py:155-212) for a batch of held-out documents.

The topic-word loadings ``ph_hat`` are fixed; every document only moves its own ``n_dk``, so documents
are independent: one lane group per document, all ``it`` sweeps inside one kernel launch
(``llda_foldin``, include/llda_gibbs.h).
"""
import numpy as np
import torch

from . import _native
from .corpus import csr_from_doc_tups
from .layout import group_layout

TEST_STREAM = 0x7E57        # default RNG stream id of test-time draws


def normalised_loadings(ph_hat):
    """columns of ph_hat divided by their sums -- the ``probs /= probs.sum(axis=0)`` of prep4test
    (LabeledLDA.py:162-167).  Returns (phn (K, V), bad (V,) bool): ``bad`` marks words whose column
    cannot be normalised (sum 0 or not finite -> numpy raises FloatingPointError in the reference and the
    WHOLE document falls back to the uniform 1/K)."""
    colsum = ph_hat.sum(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        phn = ph_hat / colsum
    bad = ~np.isfinite(phn).all(axis=0) | (colsum == 0)
    return phn, bad


def fold_in(ph_hat, alpha, doc_tups, it, thinning, seed, stream_id=TEST_STREAM, doc_base=0, device=None):
    """Run prep4test + ``it`` Gibbs sweeps for every document in ``doc_tups`` (doc2bow lists).
    Returns dict(th_hat (D, K) float64, n_dk (D, K) int64, z list of per-document topic arrays)."""
    _native.lib()
    if not torch.cuda.is_available():
        raise _native.NativeError("no HIP device visible: the fold-in sampler has no CPU fallback")
    dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    K, V = ph_hat.shape
    for doc in doc_tups:
        if not doc:
            raise ValueError("not enough values to unpack: a document has no in-vocabulary word")
    lay = group_layout(K)
    doc_off, word, freq = csr_from_doc_tups(doc_tups)
    D = len(doc_tups)
    phn, bad = normalised_loadings(np.asarray(ph_hat, dtype=np.float64))
    word_init = word
    if bad.any():
        rows = np.repeat(np.arange(D), np.diff(doc_off))
        doc_bad = np.zeros(D, dtype=bool)
        np.logical_or.at(doc_bad, rows, bad[word])
        word_init = np.where(doc_bad[rows], V, word).astype(np.int32)
        phn = np.where(bad, 0.0, phn)

    def to_dev(m, extra_uniform):
        out = np.zeros((V + 1, lay.KP), dtype=np.float64)
        out[:V, lay.topic_pos] = m.T
        if extra_uniform:
            out[V, lay.topic_pos] = 1 / K
        return torch.from_numpy(out).to(dev)

    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    d_off, d_word, d_winit, d_freq = t(doc_off, torch.int64), t(word, torch.int32), t(word_init, torch.int32), t(freq, torch.int32)
    d_ph, d_phn = to_dev(np.asarray(ph_hat, dtype=np.float64), False), to_dev(phn, True)
    z = torch.zeros((max(int(doc_off[-1]), 1),), dtype=torch.int32, device=dev)
    n_dk = torch.zeros((D, lay.KP), dtype=torch.int32, device=dev)
    th = torch.zeros((D, lay.KP), dtype=torch.float64, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    _native.foldin(d_off, d_word, d_winit, d_freq, d_ph, d_phn, D, V, K, alpha, it, thinning, seed, stream_id,
                   doc_base, z, n_dk, th, status)
    if int(status.item()) != 0:
        raise ValueError("pvals < 0, pvals > 1 or pvals contains NaNs")     # what numpy's multinomial reports
    tp = torch.from_numpy(lay.topic_pos.astype(np.int64)).to(dev)
    zt = torch.from_numpy(lay.pos_topic.astype(np.int64)).to(dev)[z.to(torch.int64)].cpu().numpy()
    return dict(th_hat=th[:, tp].cpu().numpy(), n_dk=n_dk[:, tp].cpu().numpy().astype(np.int64),
                z=[zt[doc_off[d]:doc_off[d + 1]] for d in range(D)], doc_off=doc_off, word=word, freq=freq)
