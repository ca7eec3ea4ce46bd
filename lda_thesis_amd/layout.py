"""Group layout: how the K topics of one document are spread over the lanes of a wavefront.

The reference normalises the K topic scores with ``np.sum`` (/root/reference/LabeledLDA.py:117,
CascadeLDA.py:413).  numpy adds a contiguous float64 vector *pairwise*: blocks ("leaves") of at
most 128 elements are reduced by 8 interleaved accumulators (element i goes to accumulator i & 7),
the accumulators are combined as ((0+1)+(2+3))+((4+5)+(6+7)), a tail of n % 8 elements is added
sequentially, and leaves are combined by a binary recursion that splits n at n/2 rounded down to a
multiple of 8.  Bit-exact parity with the reference therefore fixes the association order of the
sum.  The layout below makes that order lane-local on a 64-wide wavefront:

    leaf p (start_p, n_p), rel = k - start_p, chain j = rel & 7, row = rel >> 3
    lane g = 8*p + j     slot s = row

so that one accumulator chain is one lane walking its T slots, the 8 accumulators of a leaf are 8
neighbouring lanes (xor butterfly 1,2,4), and leaves are 8-lane groups combined by a short
butterfly schedule.  A document occupies G = 8 * P lanes (P = number of leaves rounded up to a
power of two), i.e. 64/G documents share a wavefront.  K with more than 8 leaves (some K in 969..1023, every
K > 1024; "wide" layouts, up to 64 leaves): P = leaves rounded up to a multiple of 8 and the G = 8 * P lanes are
NT = P / 8 "tiers" of ONE wavefront -- virtual lane gv = 64 * tier + physical lane; everything else (positions, masks,
draw order) is the same formula with the larger G, and the leaf totals are combined by walking numpy's recursion tree
(``comb``: (dst leaf, src leaf) per internal node in post-order).  All per-topic device arrays (rows of n_kw,
rows of n_dk, n_k) have row length KP = G*T; the MEMORY position of (lane g, slot s) in a row is

    pos = ((s // 4) * G + g) * 4 + s % 4      (T a multiple of 4)         pos = g*T + s   (T = 1, 2)

-- the 16-byte chunk s // 4 of all lanes is contiguous, so each of the T/4 vector loads a wavefront issues for a
row reads one contiguous run of it (with lane-major rows every instruction touched every cache line of the row).
The DRAW order of the keyed draw (oracle/llda_oracle.py draw_keyed) is the (lane, slot) order -- ``draw_rank`` --
not the memory order.
"""
import numpy as np

PW_BLOCK = 128          # numpy's pairwise-sum block size
MAX_K = 7688            # every K up to here splits into <= 64 leaves (llda_gibbs.h LLDA_MAX_K)
MAX_NARROW_LEAVES = 8   # up to 8 leaves a document is one lane group of <= 64 lanes x <= 16 slots ("narrow" layouts)
MAX_ROUNDS = 4          # depth of the leaf-combine schedule handed to the narrow kernels


def _leaves(n, start=0):
    if n <= PW_BLOCK:
        return [(start, n)]
    n2 = n // 2
    n2 -= n2 % 8
    return _leaves(n2, start) + _leaves(n - n2, start + n2)


def _tree(n, first=0):
    """numpy's recursion over leaf indices: an int (leaf) or a pair (left, right)."""
    if n <= PW_BLOCK:
        return first, 1
    n2 = n // 2
    n2 -= n2 % 8
    left, nl = _tree(n2, first)
    right, nr = _tree(n - n2, first + nl)
    return (left, right), nl + nr


def _depth(t):
    return 0 if isinstance(t, int) else 1 + max(_depth(t[0]), _depth(t[1]))


def _members(t):
    return [t] if isinstance(t, int) else _members(t[0]) + _members(t[1])


class GroupLayout(object):
    """Layout of K topics.  Attributes: K, leaves, P, G (lanes per document), T (slots per lane),
    KP (padded row length), tail (n % 8 of the last leaf), tail_row, topic_pos[K] (topic -> device
    position), pos_topic[KP] (device position -> topic, -1 in the padding), pos_lane / pos_slot[KP] and
    topic_lane / topic_slot[K] (which lane and slot hold a position / topic), draw_rank[KP] (position -> its place
    in the draw order), lm_pos[KP] (lane*T + slot -> position), rounds (leaf-combine
    schedule: ``n_rounds`` arrays of 8 partner-leaf ids, identity where a leaf idles)."""

    def __init__(self, K):
        K = int(K)
        if K < 1 or K > MAX_K:
            raise ValueError("K must be in 1..%d, got %d" % (MAX_K, K))
        self.K = K
        self.leaves = _leaves(K)
        m = len(self.leaves)
        self.wide = m > MAX_NARROW_LEAVES
        if self.wide:
            P = (m + 7) // 8 * 8
        else:
            P = 1
            while P < m:
                P *= 2
        self.m, self.P, self.G = m, P, 8 * P
        self.NT = P // 8 if self.wide else 0        # 64-lane tiers of a wide layout
        t_used = max((n + 7) // 8 for _, n in self.leaves)
        T = t_used
        if T > 2:
            T = (T + 3) // 4 * 4
        self.T_used, self.T = t_used, T
        self.KP = self.G * T
        self.rows = [n // 8 for _, n in self.leaves]
        self.tail = self.leaves[-1][1] % 8
        self.tail_row = self.leaves[-1][1] // 8
        for _, n in self.leaves[:-1]:
            assert n % 8 == 0
        self.W = W = 4 if T % 4 == 0 else T          # slots of one lane that are contiguous in memory
        g_all, s_all = np.divmod(np.arange(self.KP), T)
        self.lm_pos = (((s_all // W) * self.G + g_all) * W + s_all % W).astype(np.int32)   # lane-major index -> position
        self.pos_lane = np.zeros(self.KP, dtype=np.int32)
        self.pos_slot = np.zeros(self.KP, dtype=np.int32)
        self.pos_lane[self.lm_pos], self.pos_slot[self.lm_pos] = g_all, s_all
        self.draw_rank = (self.pos_lane * T + self.pos_slot).astype(np.int32)               # position -> draw order
        self.topic_pos = np.zeros(K, dtype=np.int32)
        self.pos_topic = np.full(self.KP, -1, dtype=np.int32)
        self.topic_lane = np.zeros(K, dtype=np.int32)
        self.topic_slot = np.zeros(K, dtype=np.int32)
        for p, (st, n) in enumerate(self.leaves):
            for rel in range(n):
                g, sl = 8 * p + (rel & 7), rel >> 3
                pos = int(self.lm_pos[g * T + sl])
                self.topic_pos[st + rel] = pos
                self.pos_topic[pos] = st + rel
                self.topic_lane[st + rel], self.topic_slot[st + rel] = g, sl
        # lane-major numbering (position = lane*T + slot): the fold-in kernel keeps its own arrays that way
        self.lm_topic_pos = (self.topic_lane * T + self.topic_slot).astype(np.int32)
        self.lm_pos_topic = np.full(self.KP, -1, dtype=np.int32)
        self.lm_pos_topic[self.lm_topic_pos] = np.arange(K, dtype=np.int32)
        self.leaf_start = np.zeros(max(8, P), dtype=np.int32)
        self.leaf_rows = np.zeros(max(8, P), dtype=np.int32)
        for p, (st, n) in enumerate(self.leaves):
            self.leaf_start[p] = st
            self.leaf_rows[p] = n // 8
        self.comb = self._post_order()
        if self.wide:
            self.rounds, self.n_rounds = [], 0
        else:
            self.rounds = self._schedule()
            self.n_rounds = len(self.rounds)
            if self.n_rounds > MAX_ROUNDS:
                raise ValueError("leaf-combine schedule deeper than %d" % MAX_ROUNDS)

    def _post_order(self):
        """numpy's recursion as a list of in-place adds over the leaf totals: for every internal node, in post-order,
        total[first leaf of its left child] += total[first leaf of its right child]; total[0] ends up as the sum."""
        tree, _ = _tree(self.K)
        out = []

        def visit(t):
            if isinstance(t, int):
                return t
            a, b = visit(t[0]), visit(t[1])
            out.append((a, b))
            return a

        visit(tree)
        return out

    def _schedule(self):
        tree, _ = _tree(self.K)
        rounds = []

        def visit(t):
            if isinstance(t, int):
                return
            visit(t[0])
            visit(t[1])
            d = _depth(t) - 1
            while len(rounds) <= d:
                rounds.append(np.arange(max(8, self.P), dtype=np.int32))
            lm, rm = _members(t[0]), _members(t[1])
            for a in lm:
                rounds[d][a] = rm[0]
            for b in rm:
                rounds[d][b] = lm[0]

        visit(tree)
        return rounds

    # ---- conversions between reference order (K) and device order (KP) ----
    def to_device(self, arr, dtype=None):
        """(..., K) array in topic order -> (..., KP) array in device order (zeros in the padding)."""
        arr = np.asarray(arr)
        out = np.zeros(arr.shape[:-1] + (self.KP,), dtype=dtype or arr.dtype)
        out[..., self.topic_pos] = arr
        return out

    def from_device(self, arr):
        """(..., KP) array in device order -> (..., K) array in topic order."""
        return np.asarray(arr)[..., self.topic_pos]

    def lane_masks(self, labs):
        """(D, K) 0/1 label matrix -> (D, G) uint16: bit s of [d, g] = label of the topic at slot s
        of lane g (0 in the padding).  Wide layouts: g is the virtual lane 64 * tier + lane."""
        labs = (np.asarray(labs) != 0).astype(np.uint32)
        out = np.zeros((labs.shape[0], self.G), dtype=np.uint32)
        np.add.at(out, (slice(None), self.topic_lane), labs << self.topic_slot.astype(np.uint32))
        return out.astype(np.uint16)

    def labs_from_masks(self, masks):
        """inverse of lane_masks: (D, G) lane masks -> (D, K) 0/1."""
        bits = np.asarray(masks).astype(np.int64) & 0xFFFF
        return ((bits[:, self.topic_lane] >> self.topic_slot) & 1).astype(np.uint8)


_CACHE = {}


def group_layout(K):
    K = int(K)
    if K not in _CACHE:
        _CACHE[K] = GroupLayout(K)
    return _CACHE[K]
