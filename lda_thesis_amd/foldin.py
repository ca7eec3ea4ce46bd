"""Test-time fold-in samplers on the device (``llda_foldin``, include/llda_gibbs.h).

Held-out documents are independent of one another -- the topic-word loadings are fixed and each
document only moves its own ``n_dk`` -- so a whole batch runs as one kernel launch, one lane group per
document, all sweeps inside the kernel.  Three reference code paths map onto it:

  LabeledLDA.prep4test / run_test          /root/reference/LabeledLDA.py:155-212
  CascadeLDA.prep4test / cascade_test      /root/reference/CascadeLDA.py:186-247
  CascadeLDA.prep4test / run_test          /root/reference/CascadeLDA.py:299-344

The initial probabilities (``prep4test``) are a K x len(doc) elementwise normalisation: LabeledLDA's are computed on
the device (``fold_in``), CascadeLDA's per-document ones on the host with the reference's own numpy expressions; both
are handed to the kernel as rows.
"""
import zlib

import numpy as np
import torch

from . import _native
from .corpus import csr_from_doc_tups
from .layout import group_layout

TEST_STREAM = 0x7E57            # default RNG stream id of LabeledLDA test-time draws
CASCADE_STREAM = 0xC0DE0000     # + K_global * level + label id of the first label of the call


def doc_key(doc_tup):
    """RNG document id of a held-out document for APIs that get no index (test_down_tree(doc, ...)):
    CRC-32 of its (ids, freqs)."""
    ids, freqs = zip(*doc_tup)
    return zlib.crc32(np.asarray(list(ids) + list(freqs), dtype=np.int32).tobytes())


EXACT_ONLY = False      # test hook (llda_foldin_args.exact_only): every site through the reference's pipeline


class Pending(object):
    """An enqueued llda_foldin launch; ``result()`` waits for it (on its stream) and brings the outputs to the host."""

    def __init__(self, stream, lay, doc_off, z, n_dk, th, status, keep):
        self.stream, self.lay, self.doc_off = stream, lay, doc_off
        self.z, self.n_dk, self.th, self.status = z, n_dk, th, status
        self._keep = keep                          # inputs of the launch stay alive until it has run

    def result(self):
        with torch.cuda.stream(self.stream):
            if int(self.status.item()) != 0:       # (synchronises with the launch)
                raise ValueError("pvals < 0, pvals > 1 or pvals contains NaNs")     # numpy.random.multinomial's complaint
            lay, dev, doc_off = self.lay, self.th.device, self.doc_off
            # (the fold-in kernel keeps its own matrices lane-major: entry lane*T + slot; z holds row positions)
            tp = torch.from_numpy(lay.lm_topic_pos.astype(np.int64)).to(dev)
            zt = torch.from_numpy(lay.pos_topic.astype(np.int64)).to(dev)[self.z.to(torch.int64)].cpu().numpy()
            out = dict(th_hat=self.th[:, tp].cpu().numpy(), n_dk=self.n_dk[:, tp].cpu().numpy().astype(np.int64),
                       z=[zt[doc_off[d]:doc_off[d + 1]] for d in range(len(doc_off) - 1)])
        self._keep = None
        return out


def _launch(ph, init_rows, init_idx, doc_tups, *, alpha, beta, it, thinning, seed, stream_id, doc_ids, c_init,
            c_loop, beta_fallback, avg_mode, device=None, stream=None, K_true=None, hold=False):
    """ph (K, V) float64; init_rows (R, K) float64; init_idx[S] row per site.  doc_ids: RNG id per document (any
    values: the whole batch is one launch).  Enqueues on ``stream`` (default: the current one), returns Pending."""
    _native.lib()
    _native.require_device()
    dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    stream = stream if stream is not None else torch.cuda.current_stream(dev)
    K, V = (ph.shape[1], ph.shape[0]) if isinstance(ph, torch.Tensor) else ph.shape     # device form: (V, KP) rows
    if K_true is not None:
        K = K_true
    lay = group_layout(K)
    doc_off, word, freq = csr_from_doc_tups(doc_tups)
    D = len(doc_tups)

    def rows_to_dev(m):                       # (R, K) -> (R, KP) lane-major rows on the device
        if isinstance(m, torch.Tensor):       # prepared on the device already (fold_in)
            return m
        out = np.zeros((m.shape[0], lay.KP), dtype=np.float64)
        out[:, lay.lm_topic_pos] = m
        return torch.from_numpy(out).to(dev)

    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    # uploads and zero fills go through the CURRENT stream (they never queue behind an earlier fold-in kernel);
    # only the kernel runs on ``stream``, after an event on the current stream
    d_off, d_word, d_freq = t(doc_off, torch.int64), t(word, torch.int32), t(freq, torch.int32)
    d_idx = t(np.asarray(init_idx, dtype=np.int32), torch.int32)
    d_ph = rows_to_dev(ph if isinstance(ph, torch.Tensor) else np.ascontiguousarray(ph.T))
    d_init = rows_to_dev(init_rows)
    valid = t((lay.lm_pos_topic >= 0).astype(np.uint8), torch.uint8)
    z = torch.zeros((max(int(doc_off[-1]), 1),), dtype=torch.int32, device=dev)
    n_dk = torch.zeros((D, lay.KP), dtype=torch.int32, device=dev)
    th = torch.zeros((D, lay.KP), dtype=torch.float64, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    d_ids = t(np.asarray(doc_ids, dtype=np.int64), torch.int64)   # every document carries its own RNG id

    def go():
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            _native.foldin(doc_off=d_off, n_dk=n_dk, th=th, D=D, doc_ids=d_ids, word=d_word, init_idx=d_idx, freq=d_freq,
                           ph=d_ph, init_rows=d_init, slot_valid=valid, z=z, status=status, K=K, iters=it,
                           thinning=thinning, alpha=alpha, beta=beta, c_init=c_init, c_loop=c_loop, seed=seed,
                           stream_id=stream_id, beta_fallback=beta_fallback, avg_mode=avg_mode, exact_only=EXACT_ONLY)
        return Pending(stream, lay, doc_off, z, n_dk, th, status, (d_off, d_word, d_freq, d_idx, d_ph, d_init, valid, d_ids))
    # hold=True: everything is uploaded and prepared, the kernels are enqueued when the caller calls the returned
    # function -- a synchronous host-to-device copy waits for kernels that are already running, so a caller with many
    # launches (CascadeLDA.test_down_tree_batch) uploads for all of them first
    return go if hold else go()


# ------------------------------------------------------------------------------------------------
# LabeledLDA
# ------------------------------------------------------------------------------------------------
def fold_in(ph_hat, alpha, doc_tups, it, thinning, seed, stream_id=TEST_STREAM, doc_base=0, device=None):
    """LabeledLDA.prep4test + run_test for a batch of doc2bow lists.  ``ph_hat``: (K, V) float64, numpy or a torch
    tensor already on the device (the running mean run_training left there).  Returns dict(th_hat (D, K),
    n_dk (D, K), z).

    prep4test's column normalisation (LabeledLDA.py:159-167: ``probs = ph_hat[:, doc]; probs /= probs.sum(axis=0)``,
    uniform 1/K for a document that holds a column which cannot be normalised) runs on the device: the fancy-indexed
    ``ph_hat[:, doc]`` is F-contiguous, so numpy reduces every column with its pairwise sum (``numpy_column_sums``
    restates that order); the division is IEEE on both sides."""
    _native.lib()
    _native.require_device()
    dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
    ph = ph_hat.to(device=dev, dtype=torch.float64) if isinstance(ph_hat, torch.Tensor) else \
        torch.from_numpy(np.ascontiguousarray(ph_hat, dtype=np.float64)).to(dev)
    K, V = ph.shape
    for doc in doc_tups:
        if not doc:
            raise ValueError("not enough values to unpack: a document has no in-vocabulary word")
    lay = group_layout(K)
    colsum = numpy_column_sums(ph)
    phn = ph / colsum                                                # 0/0 -> nan, x/0 -> inf as numpy (errstate ignore)
    bad = ~torch.isfinite(phn).all(dim=0) | (colsum == 0)
    lm = torch.from_numpy(lay.lm_topic_pos.astype(np.int64)).to(dev)
    d_ph = torch.zeros((V, lay.KP), dtype=torch.float64, device=dev)
    d_ph[:, lm] = ph.t()
    d_init = torch.zeros((V + 1, lay.KP), dtype=torch.float64, device=dev)
    d_init[:V, lm] = torch.where(bad[None, :], torch.zeros((), dtype=torch.float64, device=dev), phn).t()
    d_init[V, lm] = 1 / K                                            # row V = uniform
    doc_off, word, _ = csr_from_doc_tups(doc_tups)
    init_idx = word.copy()
    if bool(bad.any().item()):
        bad_h = bad.cpu().numpy()
        site_doc = np.repeat(np.arange(len(doc_tups)), np.diff(doc_off))
        doc_bad = np.zeros(len(doc_tups), dtype=bool)
        np.logical_or.at(doc_bad, site_doc, bad_h[word])
        init_idx = np.where(doc_bad[site_doc], V, word)
    return _launch(d_ph, d_init, init_idx, doc_tups, alpha=alpha, beta=0.0, it=it, thinning=thinning, seed=seed,
                   stream_id=stream_id, doc_ids=np.arange(len(doc_tups)) + doc_base, c_init=1.0000000005,
                   c_loop=1.0000005, beta_fallback=False, avg_mode=0, device=dev, K_true=K).result()


# ------------------------------------------------------------------------------------------------
# CascadeLDA
# ------------------------------------------------------------------------------------------------
def cascade_init_rows(ph, beta, doc_tups):
    """CascadeLDA.prep4test (CascadeLDA.py:193-198) for every document: smoothed, column-normalised
    loadings with the first ('generic') row overwritten by 1/len(doc).  Returns (rows (S, K), idx)."""
    out = []
    for tup in doc_tups:
        ids = [v for v, _ in tup]
        probs = ph[:, ids]
        probs += beta
        probs /= probs.sum(axis=0)
        probs[0, :] = 1 / len(ids)
        out.append(probs.T)
    rows = np.vstack(out) if out else np.zeros((0, ph.shape[0]))
    return rows, np.arange(rows.shape[0])


def numpy_column_sums(m):
    """``a.sum(axis=0)`` of numpy for an F-contiguous (K, n) float64 matrix -- what ``ph[:, ids]`` is -- on a torch
    tensor ``m`` (K, n): numpy reduces every column (contiguous, length K) with its pairwise sum: fewer than 8 terms
    sequentially; up to 128 terms eight strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the
    tail sequentially; beyond that split at n/2 rounded down to a multiple of 8 (the same
    order the sweep kernels use for the score sum).  Bit-identical to numpy: IEEE fp64 additions in the same order."""
    K = m.shape[0]
    if K < 8:
        res = m[0].clone()
        for k in range(1, K):
            res += m[k]
        return res
    if K <= 128:
        nb = K - K % 8
        r = m[0:8].clone()
        for i in range(8, nb, 8):
            r += m[i:i + 8]
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for k in range(nb, K):
            res = res + m[k]
        return res
    n2 = K // 2
    n2 -= n2 % 8
    return numpy_column_sums(m[:n2]) + numpy_column_sums(m[n2:])


def cascade_init_rows_device(ph, beta, doc_off, word_dev, lay):
    """cascade_init_rows on the device for loadings ``ph`` (K, V) that live there: the same elementwise expressions
    (CascadeLDA.py:193-198); ``probs.sum(axis=0)`` of the F-contiguous (K, n) matrix in numpy's pairwise order
    (``numpy_column_sums``).  -> (S, KP) lane-major rows, row d bitwise what ``cascade_init_rows`` gives."""
    probs = ph[:, word_dev] + beta                                   # (K, S): probs = ph[:, ids]; probs += beta
    probs /= numpy_column_sums(probs)
    lens = np.diff(doc_off)
    inv_len = np.repeat(1 / lens.astype(np.float64), lens)           # probs[0, :] = 1 / len(ids)
    probs[0] = torch.from_numpy(inv_len).to(ph.device)
    rows = torch.zeros((probs.shape[1], lay.KP), dtype=torch.float64, device=ph.device)
    rows[:, torch.from_numpy(lay.lm_topic_pos.astype(np.int64)).to(ph.device)] = probs.t()
    return rows


def cascade_fold_in(ph, alpha, beta, doc_tups, it, thinning, seed, stream_id, doc_ids, flat=False, device=None,
                    stream=None, defer=False, hold=False):
    """CascadeLDA.cascade_test (flat=False) / CascadeLDA.run_test (flat=True) for a batch of documents
    against the label subset whose loadings are ``ph`` (K_sub, V) -- numpy, or a torch tensor on the device (then
    the preparation of CascadeLDA.prep4test runs there too and nothing but the documents is uploaded).
    defer=True: enqueue on ``stream`` and return the Pending launch (``.result()`` gives the dict) so that launches
    of different label subsets overlap; hold=True (with defer): upload and prepare only, and return a function that
    enqueues the launch and gives the Pending."""
    for doc in doc_tups:
        if not doc:
            raise ValueError("not enough values to unpack: a document has no in-vocabulary word")
    kw = dict(alpha=alpha, beta=beta, it=it, thinning=thinning, seed=seed, stream_id=stream_id, doc_ids=doc_ids,
              c_init=1.0000005, c_loop=1.000005, beta_fallback=not flat, avg_mode=1 if flat else 0, stream=stream,
              hold=hold)
    if isinstance(ph, torch.Tensor):
        K, V = ph.shape
        lay = group_layout(K)
        doc_off, word, _ = csr_from_doc_tups(doc_tups)
        w_dev = torch.from_numpy(word.astype(np.int64)).to(ph.device)
        rows = cascade_init_rows_device(ph, beta, doc_off, w_dev, lay)
        d_ph = torch.zeros((V, lay.KP), dtype=torch.float64, device=ph.device)
        d_ph[:, torch.from_numpy(lay.lm_topic_pos.astype(np.int64)).to(ph.device)] = ph.t()
        pending = _launch(d_ph, rows, np.arange(rows.shape[0]), doc_tups, device=ph.device, K_true=K, **kw)
    else:
        ph = np.ascontiguousarray(ph, dtype=np.float64)
        rows, idx = cascade_init_rows(ph, beta, doc_tups)
        pending = _launch(ph, rows, idx, doc_tups, device=device, **kw)
    return pending if (defer or hold) else pending.result()


def cascade_fold_in_many(jobs, alpha, beta, it, thinning, seed, stream=None):
    """cascade_test for SEVERAL label subsets of the same size in ONE llda_foldin launch (the nodes of one level of
    CascadeLDA.test_down_tree whose label lists have the same length): jobs = [(ph (K, V) device tensor, doc_tups,
    stream_id, doc_ids), ...].  Every document carries the offset of its node's loadings and its node's RNG stream
    (llda_foldin_args.ph_base / doc_stream), so the result of each job equals cascade_fold_in of that job alone.
    Returns a function that waits for the launch and gives the list of th_hat arrays, one per job.  (ROCm runs only a
    few hardware queues at a time: a hundred small launches on as many streams mostly queue behind one another.)"""
    ph0 = jobs[0][0]
    dev, (K, V) = ph0.device, ph0.shape
    lay = group_layout(K)
    lm = torch.from_numpy(lay.lm_topic_pos.astype(np.int64)).to(dev)
    offs, words, freqs, ids, streams, bases, rows, phs, n_docs = [0], [], [], [], [], [], [], [], []
    for j, (ph, doc_tups, stream_id, doc_ids) in enumerate(jobs):
        for doc in doc_tups:
            if not doc:
                raise ValueError("not enough values to unpack: a document has no in-vocabulary word")
        doc_off, word, freq = csr_from_doc_tups(doc_tups)
        offs.append(doc_off[1:] + offs[-1][-1] if isinstance(offs[-1], np.ndarray) else doc_off[1:])
        words.append(word); freqs.append(freq)
        ids.append(np.asarray(doc_ids, dtype=np.int64))
        streams.append(np.full(len(doc_tups), stream_id & 0xFFFFFFFF, dtype=np.int64))
        bases.append(np.full(len(doc_tups), j * V * lay.KP, dtype=np.int64))
        n_docs.append(len(doc_tups))
        w_dev = torch.from_numpy(word.astype(np.int64)).to(dev)
        rows.append(cascade_init_rows_device(ph, beta, doc_off, w_dev, lay))
        d_ph = torch.zeros((V, lay.KP), dtype=torch.float64, device=dev)
        d_ph[:, lm] = ph.t()
        phs.append(d_ph)
    doc_off = np.concatenate([np.zeros(1, dtype=np.int64)] + [np.asarray(o, dtype=np.int64) for o in offs[1:]])
    word, freq = np.concatenate(words), np.concatenate(freqs)
    D, S = int(sum(n_docs)), int(word.shape[0])
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dt)
    d_off, d_word, d_freq = t(doc_off, torch.int64), t(word, torch.int32), t(freq, torch.int32)
    d_idx = torch.arange(S, dtype=torch.int32, device=dev)
    d_ids, d_base = t(np.concatenate(ids), torch.int64), t(np.concatenate(bases), torch.int64)
    d_stream = t(np.concatenate(streams), torch.int64).to(torch.int32)      # (uint32 bit patterns)
    d_ph, d_init = torch.cat(phs), torch.cat(rows)
    valid = t((lay.lm_pos_topic >= 0).astype(np.uint8), torch.uint8)
    z = torch.zeros((max(S, 1),), dtype=torch.int32, device=dev)
    n_dk = torch.zeros((D, lay.KP), dtype=torch.int32, device=dev)
    th = torch.zeros((D, lay.KP), dtype=torch.float64, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    stream = stream if stream is not None else torch.cuda.current_stream(dev)

    def go():
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            _native.foldin(doc_off=d_off, n_dk=n_dk, th=th, D=D, doc_ids=d_ids, word=d_word, init_idx=d_idx, freq=d_freq,
                           ph=d_ph, init_rows=d_init, slot_valid=valid, z=z, status=status, K=K, iters=it,
                           thinning=thinning, alpha=alpha, beta=beta, c_init=1.0000005, c_loop=1.000005, seed=seed,
                           stream_id=0, beta_fallback=True, avg_mode=0, ph_base=d_base, doc_stream=d_stream,
                           exact_only=EXACT_ONLY)
        keep = [(d_off, d_word, d_freq, d_idx, d_ids, d_base, d_stream, d_ph, d_init, valid)]   # alive until it has run

        def result():
            with torch.cuda.stream(stream):
                if int(status.item()) != 0:
                    raise ValueError("pvals < 0, pvals > 1 or pvals contains NaNs")
                tp = torch.from_numpy(lay.lm_topic_pos.astype(np.int64)).to(dev)
                allth = th[:, tp].cpu().numpy()
            keep.clear()
            bounds = np.concatenate(([0], np.cumsum(n_docs)))
            return [allth[bounds[j]:bounds[j + 1]] for j in range(len(jobs))]
        return result
    return go
