"""Adversarial near-tie inputs for the tiered draw (test infrastructure).

The HIP sweep decides most sites in fp32 (tier 0): it is allowed to do so only when the keyed
threshold t = u * total is further than a margin (2^-17 of the total) from every prefix sum of the topic scores,
because the fp32 evaluation is only proven to be within 105 * 2^-24 of the real value.  A site whose threshold lies
within ~2^-26 of a prefix sum is the worst case for that scheme: the sign of (prefix - t) -- i.e. the chosen topic --
is decided far below fp32 resolution, so tier 0 MUST notice that it cannot decide; if its rounding error ever
exceeded the margin it would be "sure" about such a site and pick the wrong topic half of the time.

This module builds a state (reference layout: n_d_k, n_k_v, n_zk, z) in which the LAST site of every document is
such a near tie and all earlier sites are comfortably decided (so the state the tuned site sees is known).  Every
site has its own word, so its n_k_v column is a free parameter: the column is random (heavy tailed, wide dynamic
range: the case the fp32 prefix sums like least) and then one or two of its entries are nudged by whole counts until
the threshold sits on a prefix-sum boundary.  The scores follow /root/reference/LabeledLDA.py:113-118, the draw
oracle/llda_oracle.py:draw_keyed (device position order).
"""
import numpy as np

import llda_oracle as orc


def _site_scores(lay, lab, nd, nk, x, alpha, beta, vbeta):
    """unnormalised scores in device position order (padding = 0) and their inclusive prefix sums."""
    w = lab * (nd + alpha) * ((x + beta) / (nk + vbeta))
    wp = np.zeros(lay.KP)
    wp[lay.topic_slot] = w
    return wp, np.cumsum(wp)


def _gap(cum, t):
    """distance of t to the nearest prefix sum, relative to the total."""
    return float(np.min(np.abs(cum - t)) / cum[-1])


def make_neartie_state(K, D, dense, seed, rng, max_sites=4, safe=2.0 ** -14, tuned=2.0 ** -24, alpha=0.1, beta=0.01,
                       stream=0, doc_base=0, label_count=None):
    """-> dict(doc_off, word, freq, z, labs, n_d_k, n_k_v, n_zk, V, n_tuned, tuned_gap_max, safe_gap_min)."""
    lay = orc.layout(K)
    lens = rng.integers(1, max_sites + 1, size=D)
    doc_off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    S = int(doc_off[-1])
    V = S                                                    # every site has its own word
    vbeta = V * beta
    word = np.arange(S, dtype=np.int32)
    freq = rng.integers(1, 4, size=S).astype(np.int32)
    n_zk = np.floor(np.exp(rng.uniform(np.log(1e3), np.log(1e6), size=K))).astype(np.int64)
    labs = np.ones((D, K), dtype=np.uint8)
    if not dense and label_count is not None:                # sparse label sets: root + up to label_count labels per document
        labs = np.zeros((D, K), dtype=np.uint8)
        labs[:, 0] = 1
        for d in range(D):
            n = int(rng.integers(max(1, label_count - 2), label_count + 1))
            labs[d, rng.choice(K - 1, size=min(n, K - 1), replace=False) + 1] = 1
    elif not dense:
        labs = (rng.random((D, K)) < 0.4).astype(np.uint8)
        labs[:, 0] = 1
    z = np.zeros(S, dtype=np.int64)
    n_d_k = np.zeros((D, K), dtype=np.int64)
    n_k_v = np.zeros((K, V), dtype=np.int64)
    slot_of = lay.topic_slot                                 # topic -> device position
    n_tuned, tuned_gap_max, safe_gap_min = 0, 0.0, 1.0
    for d in range(D):
        allowed = np.flatnonzero(labs[d])
        lab = labs[d].astype(np.float64)
        s0, L = int(doc_off[d]), int(lens[d])
        zo = rng.choice(allowed, size=L)
        z[s0:s0 + L] = zo
        nd = np.where(rng.random(K) < 0.1, rng.integers(0, 2000, size=K), rng.integers(0, 20, size=K)) * labs[d]
        np.add.at(nd, zo, freq[s0:s0 + L])
        n_d_k[d] = nd
        nd = nd.astype(np.float64)
        nk = n_zk.astype(np.float64)
        us = orc.keyed_uniform(seed, 0, stream, d + doc_base, np.arange(L))
        for n in range(L):
            i, f, u = s0 + n, float(freq[s0 + n]), float(us[n])
            nd[zo[n]] -= f
            nk[zo[n]] -= f
            last = n == L - 1
            for _attempt in range(200):
                x = np.floor(np.exp(rng.uniform(0.0, 11.5, size=K)))          # 1 .. 1e5, heavy tailed
                x[rng.random(K) < 0.3] = 0.0
                if last:
                    ok = _tune(lay, lab, nd, nk, x, u, alpha, beta, vbeta, allowed, slot_of, tuned)
                    if not ok:
                        continue
                wp, cum = _site_scores(lay, lab, nd, nk, x, alpha, beta, vbeta)
                gap = _gap(cum, u * cum[-1])
                if last or gap > safe:
                    break
            else:
                raise RuntimeError("could not place site %d of document %d" % (n, d))
            if last:
                n_tuned += 1
                tuned_gap_max = max(tuned_gap_max, gap)
            else:
                safe_gap_min = min(safe_gap_min, gap)
            col = x.astype(np.int64)
            col[zo[n]] += int(f)                             # the stored column includes the site's own count
            n_k_v[:, i] = col
            # the comfortably decided draw (never used for the tuned site: it is the document's last)
            zn = orc.draw_keyed((wp / cum[-1])[slot_of], u, lay)
            nd[zn] += f
            nk[zn] += f
    return dict(doc_off=doc_off, word=word, freq=freq, z=z, labs=labs, n_d_k=n_d_k, n_k_v=n_k_v, n_zk=n_zk, V=V,
                alpha=alpha, beta=beta, n_tuned=n_tuned, tuned_gap_max=tuned_gap_max, safe_gap_min=safe_gap_min)


def _tune(lay, lab, nd, nk, x, u, alpha, beta, vbeta, allowed, slot_of, target):
    """nudge entries of the column x (in place, whole counts, >= 0) until u * total is within `target` of a prefix
    sum.  Raising x[j] by one raises the score of topic j by c_j = (nd_j + alpha) / (nk_j + V beta): a topic behind
    the boundary moves only the threshold (by u c_j), a topic before it moves the boundary too."""
    c = lab * (nd + alpha) / (nk + vbeta)
    for _round in range(6):
        wp, cum = _site_scores(lay, lab, nd, nk, x, alpha, beta, vbeta)
        t = u * cum[-1]
        b = int(np.searchsorted(cum, t, side="right"))       # first boundary above t
        if b >= lay.KP:
            return False
        E = cum[b] - t                                       # > 0: make it ~0
        if E / cum[-1] < target:
            return True
        behind = [j for j in allowed if slot_of[j] > b and c[j] > 0]
        before = [j for j in allowed if slot_of[j] < b and c[j] > 0]
        best = None
        for j in behind:                                     # x[j] += k lowers E by k u c_j
            k = np.floor(E / (u * c[j]))
            if k >= 1 and x[j] + k < 2e8:
                r = E - k * u * c[j]
                if best is None or r < best[0]:
                    best = (r, j, k)
        for j in before:                                     # x[j] -= k lowers E by k (1 - u) c_j
            k = min(np.floor(E / ((1 - u) * c[j])), x[j])
            if k >= 1:
                r = E - k * (1 - u) * c[j]
                if best is None or r < best[0]:
                    best = (r, j, -k)
        if best is None:
            return False
        x[best[1]] += best[2]
    wp, cum = _site_scores(lay, lab, nd, nk, x, alpha, beta, vbeta)
    return _gap(cum, u * cum[-1]) < target
