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


PLACES = ("lane0", "lastlane", "seam", "last")


def make_neartie_state(K, D, dense, seed, rng, max_sites=4, safe=2.0 ** -14, tuned=2.0 ** -24, alpha=0.1, beta=0.01,
                       stream=0, doc_base=0, label_count=None, xcap=2e8, xlog=11.5, nd_big=2000, place=None, wide_every=0,
                       rounds=6):
    """-> dict(doc_off, word, freq, z, labs, n_d_k, n_k_v, n_zk, V, n_tuned, tuned_gap_max, safe_gap_min, ...).

    xcap, xlog : every entry of n_k_v stays below xcap (65 000: the whole state fits the 16-bit image of the quad kernels), the random
                 columns are exp(uniform(0, xlog)); nd_big: range of the large entries of n_d_k (documents below 2^16 tokens: 600).
    place      : None = the boundary the keyed uniform happens to select; one of PLACES, or "mix" (documents take PLACES in turn) =
                 the column is shaped so that the tuned boundary is where the quad kernel's tier 0 has the least room
                 (csrc/kernel_quad.hpp: a document is G / 2 quad lanes, each walking standard lanes 2 lq (chain A) and 2 lq + 1 (B)):
                 "lane0"    a slot of quad lane 0 -- no lanes before it, the data-dependent margin is at its smallest;
                 "lastlane" a slot of the last quad lane -- the scan's longest sum in front of it;
                 "seam"     the last slot of a standard lane -- chain A's total against chain B's first element, or the lane total;
                 "last"     the last but one position with a topic -- everything behind the boundary is ONE slot (the key "no slot
                            above the threshold" names it).
                 -> place_of[D] (index into PLACES, -1 = not placed: the uniform did not allow it) and tuned_pos[D] (draw-order
                 position of the boundary).
    wide_every : n > 0 = in every n-th document the tuned site's word has counts beyond 65 535 (one of them beyond 2^24): the quad
                 kernel reads such a row as int32, without tier 1 (xcap does not apply to those entries).
    """
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
    place_of = np.full(D, -1, dtype=np.int64)
    tuned_pos = np.full(D, -1, dtype=np.int64)
    tuned_gap = np.zeros(D)
    tuned_terms = np.zeros((D, 4))                           # (lane total, u * total, lanes before, total) of the tuned site, quad lanes
    wide_docs = np.zeros(D, dtype=bool)
    for d in range(D):
        allowed = np.flatnonzero(labs[d])
        lab = labs[d].astype(np.float64)
        s0, L = int(doc_off[d]), int(lens[d])
        zo = rng.choice(allowed, size=L)
        z[s0:s0 + L] = zo
        nd = np.where(rng.random(K) < 0.1, rng.integers(0, nd_big, size=K), rng.integers(0, 20, size=K)) * labs[d]
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
            want = -1
            if last and place is not None:
                want = d % len(PLACES) if place == "mix" else PLACES.index(place)
            wide = bool(last and wide_every and d % wide_every == 0)
            for _attempt in range(400):
                x = np.minimum(np.floor(np.exp(rng.uniform(0.0, xlog, size=K))), xcap - 8)      # 1 .. 1e5, heavy tailed
                x[rng.random(K) < 0.3] = 0.0
                frozen = None
                if wide:                                     # counts the 16-bit image cannot hold; the tuner leaves them alone
                    frozen = rng.choice(allowed, size=min(3, len(allowed)), replace=False)
                    x[frozen] = np.floor(np.exp(rng.uniform(np.log(7e4), np.log(4e5), size=len(frozen))))
                    x[frozen[0]] = float(rng.integers(1 << 24, 1 << 25))
                placed = False
                if last:
                    if want >= 0 and _attempt < 300:
                        placed = _place(lay, lab, nd, nk, x, u, alpha, beta, vbeta, slot_of, PLACES[want], rng, xcap, frozen)
                        if not placed:
                            continue
                    ok = _tune(lay, lab, nd, nk, x, u, alpha, beta, vbeta, allowed, slot_of, tuned, xcap, frozen, rounds)
                    if not ok:
                        continue
                wp, cum = _site_scores(lay, lab, nd, nk, x, alpha, beta, vbeta)
                gap = _gap(cum, u * cum[-1])
                if last and placed and not _is_place(lay, int(np.argmin(np.abs(cum - u * cum[-1]))), PLACES[want]):
                    continue                                 # (the tuner moved the boundary somewhere else)
                if last or gap > safe:
                    break
            else:
                raise RuntimeError("could not place site %d of document %d" % (n, d))
            if last:
                n_tuned += 1
                tuned_gap_max = max(tuned_gap_max, gap)
                tuned_gap[d] = gap
                b = int(np.argmin(np.abs(cum - u * cum[-1])))
                tuned_pos[d] = b
                place_of[d] = want if placed else -1
                wide_docs[d] = wide
                if lay.G >= 2:                                # the terms of the quad kernel's data-dependent margin
                    ql = b // (2 * lay.T)
                    lane_tot = wp.reshape(lay.G // 2, 2 * lay.T).sum(axis=1)
                    tuned_terms[d] = (lane_tot[ql], u * cum[-1], lane_tot[:ql].sum(), cum[-1])
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
                alpha=alpha, beta=beta, n_tuned=n_tuned, tuned_gap_max=tuned_gap_max, safe_gap_min=safe_gap_min,
                place_of=place_of, tuned_pos=tuned_pos, tuned_gap=tuned_gap, tuned_terms=tuned_terms, wide_docs=wide_docs)


def _is_place(lay, b, kind):
    """is draw-order position b (lane * T + slot) a boundary of the wanted kind?"""
    T, G = lay.T, lay.G
    valid = np.flatnonzero(lay.slot_topic >= 0)
    g = b // T
    if kind == "lane0":
        return g < 2
    if kind == "lastlane":
        return g >= G - 2 and b != valid[-1]
    if kind == "seam":
        nxt = valid[np.searchsorted(valid, b, side="right")] if b != valid[-1] else -1
        return nxt >= 0 and nxt // T != g                    # the next position with a topic is in another standard lane
    return b == valid[-2]


def _place(lay, lab, nd, nk, x, u, alpha, beta, vbeta, slot_of, kind, rng, xcap, frozen):
    """shape the column x (in place) so that the first prefix sum above u * total is a boundary of the wanted kind, with the threshold
    in the upper tenth of that (wide) slot: the entries before / behind it are scaled down until the prefix share matches u."""
    T, G, KP = lay.T, lay.G, lay.KP
    valid = np.flatnonzero((lay.slot_topic >= 0) & (lab[np.maximum(lay.slot_topic, 0)] > 0))
    cand = np.array([b for b in valid if _is_place(lay, int(b), kind)])
    if len(cand) == 0:
        return False
    b = int(rng.choice(cand))
    jb = int(lay.slot_topic[b])
    hold = set() if frozen is None else set(int(j) for j in frozen)
    if jb in hold:
        return False
    cap = min(xcap, 1e5)
    x[jb] = np.floor(rng.uniform(0.3, 0.9) * cap)            # a wide slot: room for the tuner behind the threshold
    theta = 0.1
    if kind == "last":
        # everything behind the boundary is ONE slot: make that one wide, and the boundary's slot as wide as the uniform allows
        # (0.9 Wb < u (Wb + B0), or no scaling of what lies before can bring the threshold into the slot)
        jl = int(lay.slot_topic[valid[-1]])
        if jl in hold:
            return False
        x[jl] = np.floor(rng.uniform(0.5, 0.95) * cap)
        if u < 0.85:
            c = lab * (nd + alpha) / (nk + vbeta)
            room = 0.5 * u * c[jl] * (x[jl] + beta) / (0.9 - u)
            x[jb] = min(x[jb], np.floor(room / c[jb] - beta))
            if x[jb] < 1:
                return False
    pos = np.asarray(slot_of)                                # topic -> draw-order position
    keep = np.zeros(len(x), dtype=bool)
    keep[jb] = True
    keep[list(hold)] = True
    for _ in range(4):
        wp, cum = _site_scores(lay, lab, nd, nk, x, alpha, beta, vbeta)
        Wb, A0 = wp[b], cum[b] - wp[b]
        B0 = cum[-1] - cum[b]
        # A0 sA + Wb (1 - theta) = u (A0 sA + Wb + B0 sB)
        sA, sB = 1.0, 1.0
        if B0 > 0:
            sB = ((A0 + Wb * (1 - theta)) / u - A0 - Wb) / B0
        if sB > 1.0 or B0 <= 0:
            sB = 1.0
            if A0 <= 0:
                return False
            sA = (u * (Wb + B0) - Wb * (1 - theta)) / (A0 * (1 - u))
        if not (0.0 < sA <= 1.0 and 0.0 < sB <= 1.0):
            return False
        if sA < 1.0:
            m = (pos < b) & ~keep
            x[m] = np.floor(x[m] * sA)
        if sB < 1.0:
            m = (pos > b) & ~keep
            x[m] = np.floor(x[m] * sB)
        wp, cum = _site_scores(lay, lab, nd, nk, x, alpha, beta, vbeta)
        t = u * cum[-1]
        if int(np.searchsorted(cum, t, side="right")) == b and (cum[b] - t) < 0.5 * wp[b]:
            return True
    return False


def _tune(lay, lab, nd, nk, x, u, alpha, beta, vbeta, allowed, slot_of, target, xcap=2e8, frozen=None, rounds=6):
    """nudge entries of the column x (in place, whole counts, >= 0) until u * total is within `target` of a prefix
    sum.  Raising x[j] by one raises the score of topic j by c_j = (nd_j + alpha) / (nk_j + V beta): a topic behind
    the boundary moves only the threshold (by u c_j), a topic before it moves the boundary too."""
    c = lab * (nd + alpha) / (nk + vbeta)
    if frozen is not None:
        allowed = np.setdiff1d(allowed, frozen)
    for _round in range(rounds):
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
            if k >= 1 and x[j] + k < xcap:
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
