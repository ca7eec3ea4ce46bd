"""Corpus containers and synthetic corpora (CSR form the sampler eats).

The reference keeps documents as python lists of (word id, frequency) tuples produced by gensim's
``doc2bow`` (/root/reference/LabeledLDA.py:64,80-84): word ids inside a document are unique and
ascending.  Here a corpus is three flat arrays -- ``doc_off`` (D+1), ``word`` (S), ``freq`` (S).
"""
from itertools import chain

import numpy as np
import torch


def csr_from_doc_tups(doc_tups):
    """list of [(word id, freq), ...] (doc2bow output) -> (doc_off int64, word int32, freq int32)."""
    lens = np.fromiter(map(len, doc_tups), dtype=np.int64, count=len(doc_tups))
    doc_off = np.zeros(len(doc_tups) + 1, dtype=np.int64)
    np.cumsum(lens, out=doc_off[1:])
    S = int(doc_off[-1])
    flat = chain.from_iterable(chain.from_iterable(doc_tups))          # id, f, id, f, ... at C speed
    pairs = np.fromiter(flat, dtype=np.int64, count=2 * S).reshape(S, 2)
    return doc_off, pairs[:, 0].astype(np.int32), pairs[:, 1].astype(np.int32)


def cascade_corpus_from_csr(doc_off, word, freq, lab_off, lab_idx, names, depth=3):
    """Token lists and prefix-expanded label lists (what CascadeLDA.load_corpus returns, reference
    CascadeLDA.py:8-49) rebuilt from a tokenised corpus in CSR form: word id v with frequency f becomes f copies of
    the pseudo-token 'w%05d' % v, every label code (names[k], k != 0 = 'root') is expanded to its prefixes up to
    ``depth`` in first-seen order.  -> (docs, labs, labelset)."""
    docs, labs, seen = [], [], {}
    for d in range(len(doc_off) - 1):
        toks = []
        for v, f in zip(word[doc_off[d]:doc_off[d + 1]], freq[doc_off[d]:doc_off[d + 1]]):
            toks += ["w%05d" % v] * int(f)
        docs.append(toks)
        lab = []
        for k in lab_idx[lab_off[d]:lab_off[d + 1]]:
            if k != 0:
                for i in range(depth):
                    p = names[k][:i + 1]
                    if p not in lab:
                        lab.append(p)
        for x in lab:
            seen.setdefault(x, 1)
        labs.append(lab)
    return docs, labs, list(seen.keys())


def zipf_cdf(V, s=1.0, device="cpu"):
    w = 1.0 / torch.arange(1, V + 1, dtype=torch.float64, device=device) ** s
    cdf = torch.cumsum(w, 0)
    return cdf / cdf[-1].clone()


def synthetic_corpus(D, N, V, K, seed, device, chunk=32768, oversample=3, zipf_s=1.0):
    """D documents of exactly N distinct word ids each (ascending, f = 1): successive sampling
    WITHOUT replacement from Zipf(s=zipf_s) over V words (zipf_s = 0: uniform); initial topics uniform over K.
    Generated on ``device`` with a torch.Generator seeded by ``seed``.
    Returns (doc_off int64 (D+1), word int32 (D*N), freq int32, z int64 topic ids)."""
    if N > V:
        raise ValueError("cannot draw %d distinct words from a vocabulary of %d" % (N, V))
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    cdf = zipf_cdf(V, float(zipf_s), dev)
    M = min(max(oversample * N, N + 64), 64 * N)
    words = torch.empty((D, N), dtype=torch.int32, device=dev)
    for lo in range(0, D, chunk):
        hi = min(D, lo + chunk)
        n = hi - lo
        todo = torch.arange(n, device=dev)
        out = torch.empty((n, N), dtype=torch.int64, device=dev)
        m = M
        while todo.numel():
            u = torch.rand((todo.numel(), m), dtype=torch.float64, device=dev, generator=gen)
            cand = torch.searchsorted(cdf, u).clamp_(max=V - 1)
            # first occurrence of every value in draw order
            sv, si = torch.sort(cand, dim=1, stable=True)
            first = torch.ones_like(sv, dtype=torch.bool)
            first[:, 1:] = sv[:, 1:] != sv[:, :-1]
            keep = torch.zeros_like(first)
            keep.scatter_(1, si, first)
            rank = torch.cumsum(keep.to(torch.int32), dim=1)
            ok = rank[:, -1] >= N
            sel = keep & (rank <= N)
            rows_ok = torch.nonzero(ok).flatten()
            if rows_ok.numel():
                picked = cand[rows_ok][sel[rows_ok]].view(rows_ok.numel(), N)
                out[todo[rows_ok]] = torch.sort(picked, dim=1).values
            todo = todo[~ok]
            m = min(2 * m, 64 * N)
        words[lo:hi] = out.to(torch.int32)
    doc_off = torch.arange(0, D + 1, dtype=torch.int64, device=dev) * N
    word = words.reshape(-1)
    freq = torch.ones_like(word)
    z = torch.randint(0, K, (D * N,), dtype=torch.int64, device=dev, generator=gen)
    return doc_off, word, freq, z


BLOCK_DOCS = 15625      # 1 000 000 / 64: the unit in which synthetic_corpus_blocks seeds its generator


def synthetic_corpus_blocks(doc_lo, doc_hi, N, V, K, seed, device, zipf_s=1.0, block=BLOCK_DOCS):
    """Documents [doc_lo, doc_hi) of a corpus that is generated in fixed blocks of ``block`` documents, block b
    from seed + b -- so the corpus (and the initial topics) are the same whichever way the documents are later
    split over GPUs, as long as the split points are multiples of ``block``.  Returns the local CSR as
    synthetic_corpus does."""
    if doc_lo % block or doc_hi % block:
        raise ValueError("shard bounds must be multiples of %d documents" % block)
    words, zs = [], []
    for lo in range(doc_lo, doc_hi, block):
        n = block
        _, w, _, z = synthetic_corpus(n, N, V, K, seed + lo // block, device, zipf_s=zipf_s)
        words.append(w)
        zs.append(z)
    word = torch.cat(words)
    D = doc_hi - doc_lo
    doc_off = torch.arange(0, D + 1, dtype=torch.int64, device=word.device) * N
    return doc_off, word, torch.ones_like(word), torch.cat(zs)
