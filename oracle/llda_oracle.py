"""CPU oracle for the collapsed-Gibbs sweep of LabeledLDA / SubLDA  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``lda_thesis_amd``) never imports anything under ``oracle/`` and raises if the HIP
library is missing.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the unmodified reference from
``/root/reference`` (gensim stubbed by ``oracle/refshim.py``), runs the reference's own
``training_iteration`` in the three modes below and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this file (and the C restatement ``llda_oracle.c``) against
those vectors bit for bit.

Reference lines restated here (all in /root/reference):
  * sweep body           LabeledLDA.py:101-125  ==  CascadeLDA.py:397-421 (SubLDA)
  * count initialisation LabeledLDA.py:80-92, CascadeLDA.py:373-385 (with the phantom-column quirk)
  * read-outs            LabeledLDA.py:231-239 (get_phi/get_theta), :256-265 (perplexity),
                         CascadeLDA.py:394-395 (get_ph)
  * np.sum               numpy 2.2.6 pairwise_sum (third party, not vendored): restated in
                         ``pairwise_sum`` and checked against np.sum in tests/test_oracle_units.py

Three execution modes of the same per-site body:
  O1  sequential, numpy's own legacy stream (np.random.multinomial) -- the reference verbatim.
  O2  sequential, keyed draw  (``KeyedDraw``: Philox uniform keyed on sweep/doc/site).
  O3  per-document snapshot, keyed draw: every document reads the sweep-start n_k_v / n_zk plus its
      own changes; integer deltas are summed afterwards.  This is the semantics the HIP kernel
      implements; it is independent of document order and of the number of GPUs.

The keyed categorical draw (``draw_keyed``) is OURS to define (the reference takes whatever
``multinom_draw`` is bound to, LabeledLDA.py:4,119).  Its summation order is the "group layout"
below, chosen so that numpy's pairwise sum and the prefix scan are both lane-local on a wavefront.
"""
import hashlib

import numpy as np

# ----------------------------------------------------------------------------------------------
# Group layout of the K topics  (mirrors numpy's pairwise_sum blocking)
# ----------------------------------------------------------------------------------------------
PW_BLOCK = 128     # numpy PW_BLOCKSIZE
MAX_K = 7688       # every K up to here splits into <= 64 leaves (8 'tiers' of 64 lanes on the device)


def pairwise_leaves(n, start=0):
    """Leaves (start, length<=128) of numpy's pairwise_sum recursion, left to right."""
    if n <= PW_BLOCK:
        return [(start, n)]
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise_leaves(n2, start) + pairwise_leaves(n - n2, start + n2)


def _tree(n, first_leaf=0):
    """Recursion tree over leaf indices: int (a leaf) or (left, right)."""
    if n <= PW_BLOCK:
        return first_leaf, 1
    n2 = n // 2
    n2 -= n2 % 8
    l, nl = _tree(n2, first_leaf)
    r, nr = _tree(n - n2, first_leaf + nl)
    return (l, r), nl + nr


class Layout(object):
    """Topic k  <->  (lane g, slot s) of a G-lane group.

    leaf p = numpy pairwise leaf containing k, rel = k - start_p, chain j = rel & 7, row = rel >> 3
      lane g = 8*p + j,  slot s = row,  storage position pos = g*T + s   (pos is the device index)
    P = leaves rounded up to a power of two (up to 8 leaves) or to a multiple of 8 (more: "wide" layouts, the 8*P
    lanes are then 64-lane tiers of one wavefront), G = 8*P lanes, T = slots per lane (rounded up to a
    multiple of 4 when > 2), KP = G*T padded row length.  A leaf of n topics has R = n//8 full rows
    (slots 0..R-1, summed by the 8 chains) and n%8 tail topics in slot R of lanes j < n%8
    (added sequentially after the chains are combined) -- only the last leaf can have a tail.
    """

    def __init__(self, K):
        if K < 1 or K > MAX_K:
            raise ValueError("K must be in 1..%d" % MAX_K)
        self.K = K
        self.leaves = pairwise_leaves(K)
        m = len(self.leaves)
        if m > 8:
            P = (m + 7) // 8 * 8
        else:
            P = 1
            while P < m:
                P *= 2
        self.m, self.P, self.G = m, P, 8 * P
        t_used = max((n + 7) // 8 for _, n in self.leaves)
        T = t_used
        if T > 2:
            T = (T + 3) // 4 * 4
        self.T_used, self.T = t_used, T
        self.KP = self.G * T
        self.rows = [n // 8 for _, n in self.leaves]        # R_p
        self.tails = [n % 8 for _, n in self.leaves]        # only the last may be non-zero
        self.tree, _ = _tree(K)
        self.slot_topic = np.full(self.KP, -1, dtype=np.int32)
        self.topic_slot = np.zeros(K, dtype=np.int32)
        for p, (st, n) in enumerate(self.leaves):
            for rel in range(n):
                pos = (8 * p + (rel & 7)) * T + (rel >> 3)
                self.slot_topic[pos] = st + rel
                self.topic_slot[st + rel] = pos

    def grid(self, vec):
        """length-K vector -> (G, T) array in group layout, zeros in the padding."""
        out = np.zeros(self.KP, dtype=np.asarray(vec).dtype)
        out[self.topic_slot] = vec
        return out.reshape(self.G, self.T)

    def combine_rounds(self):
        """Butterfly schedule for the leaf totals: list of rounds, each a length-P array giving the
        partner part of every part (itself = idle).  x_p <- x_p + x_partner reproduces the
        recursion tree because fp addition is commutative."""
        rounds = []

        def depth(t):
            return 0 if isinstance(t, int) else 1 + max(depth(t[0]), depth(t[1]))

        def members(t):
            return [t] if isinstance(t, int) else members(t[0]) + members(t[1])

        def visit(t):
            if isinstance(t, int):
                return
            visit(t[0])
            visit(t[1])
            d = depth(t) - 1
            while len(rounds) <= d:
                rounds.append(np.arange(self.P, dtype=np.int32))
            lm, rm = members(t[0]), members(t[1])
            for a in lm:
                rounds[d][a] = rm[0]
            for b in rm:
                rounds[d][b] = lm[0]

        visit(self.tree)
        return rounds


_LAYOUTS = {}


def layout(K):
    if K not in _LAYOUTS:
        _LAYOUTS[K] = Layout(K)
    return _LAYOUTS[K]


def pairwise_sum(a):
    """np.sum of a contiguous float64 vector, restated (numpy/_core/src/umath/loops_utils.h.src)."""
    a = np.asarray(a, dtype=np.float64)
    n = a.shape[0]
    if n < 8:
        res = 0.0
        for i in range(n):
            res = res + a[i]
        return res
    if n <= PW_BLOCK:
        r = [a[j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = r[j] + a[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return pairwise_sum(a[:n2]) + pairwise_sum(a[n2:])


def group_sum(lay, w):
    """The same sum evaluated the way the wavefront does it: per-lane chains over the slots, xor
    butterfly over the 8 chains of a leaf, sequential tail, butterfly over the leaves."""
    g = lay.grid(np.asarray(w, dtype=np.float64))
    part = np.zeros(lay.P, dtype=np.float64)
    for p in range(lay.m):
        R, t = lay.rows[p], lay.tails[p]
        acc = np.zeros(8, dtype=np.float64)
        for s in range(R):
            acc = acc + g[8 * p:8 * p + 8, s]
        for sh in (1, 2, 4):
            acc = acc + acc[np.arange(8) ^ sh]
        res = acc[0]
        for j in range(t):
            res = res + g[8 * p + j, R]
        part[p] = res
    for partner in lay.combine_rounds():
        part = np.where(partner == np.arange(lay.P), part, part + part[partner])
    return part[0]


# ----------------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
# ----------------------------------------------------------------------------------------------
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32 with 10 rounds.  Inputs broadcastable uint32-valued arrays."""
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> _S32, p0 & _MASK
        hi1, lo1 = p1 >> _S32, p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def keyed_uniform(seed, sweep, stream, doc, site):
    """53-bit uniform in [0,1) for (seed; sweep, stream, doc, site).

    One Philox block serves two consecutive sites of a document:
    counter = (site >> 1, doc, stream, sweep), key = (seed & 0xffffffff, seed >> 32);
    (a, b) = (r0, r1) for an even site, (r2, r3) for an odd site;
    u = ((a >> 5) * 2**26 + (b >> 6)) / 2**53   (all operations exact in float64).
    """
    seed = int(seed)
    site = np.asarray(site, dtype=np.uint64)
    r0, r1, r2, r3 = philox4x32_10(site >> np.uint64(1), doc, stream, sweep,
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    odd = (site & np.uint64(1)).astype(bool)
    a = (np.where(odd, r2, r0) >> np.uint64(5)).astype(np.float64)
    b = (np.where(odd, r3, r1) >> np.uint64(6)).astype(np.float64)
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)


# ----------------------------------------------------------------------------------------------
# The keyed categorical draw
# ----------------------------------------------------------------------------------------------
def draw_keyed(prob, u, lay=None):
    """Topic chosen for the normalised probability vector ``prob`` (length K) and uniform ``u``.

    In group layout: q[g][s] = sequential inclusive prefix over the slots of lane g;
    X = Hillis-Steele inclusive scan of the lane totals q[g][T-1] over the G lanes
    (for d = 1,2,4,..: x[g] <- x[g-d] + x[g] where g >= d);  t = u * X[G-1];
    t_g = t - X[g-1]  (X[-1] := 0);  the draw is the first (g, s) in lane-major order with
    prob > 0 and q[g][s] > t_g, or the last position with prob > 0 if there is none.
    """
    prob = np.asarray(prob, dtype=np.float64)
    if lay is None:
        lay = layout(prob.shape[0])
    p = lay.grid(prob)
    q = p.copy()
    for s in range(1, lay.T):
        q[:, s] = q[:, s - 1] + p[:, s]
    x = q[:, lay.T - 1].copy()
    d = 1
    while d < lay.G:
        y = x.copy()
        y[d:] = x[:-d] + x[d:]
        x = y
        d *= 2
    t = u * x[lay.G - 1]
    off = np.concatenate(([0.0], x[:-1]))
    tg = t - off
    pos = p > 0.0
    flag = (pos & (q > tg[:, None])).ravel()
    if flag.any():
        return int(lay.slot_topic[int(np.argmax(flag))])
    pos = pos.ravel()
    if not pos.any():
        raise FloatingPointError("draw_keyed: no positive probability")
    return int(lay.slot_topic[pos.shape[0] - 1 - int(np.argmax(pos[::-1]))])


class KeyedDraw(object):
    """Callable with numpy.random.multinomial's call shape: draw(1, prob) -> one-hot int array.
    Injected into the reference as the module global ``multinom_draw`` by gen_golden.py.  The
    driver sets ``.sweep`` and either ``.doc``/``.site`` (site auto-increments) or a ``.plan``
    iterator yielding (global_doc, site) in visit order."""

    def __init__(self, seed, stream=0):
        self.seed = seed
        self.stream = stream
        self.sweep = 0
        self.doc = 0
        self.site = 0
        self.plan = None
        self.calls = 0

    def __call__(self, n, prob):
        if self.plan is not None:
            self.doc, self.site = next(self.plan)
        u = float(keyed_uniform(self.seed, self.sweep, self.stream, self.doc, self.site))
        self.site += 1
        self.calls += 1
        z = draw_keyed(prob, u)
        out = np.zeros(len(prob), dtype=np.int64)
        out[z] = 1
        return out


# ----------------------------------------------------------------------------------------------
# State container and initialisation (reference a3/a4)
# ----------------------------------------------------------------------------------------------
class State(object):
    """Reference-shaped sampler state: n_d_k (D,K) int64, n_k_v (K,V) int64, n_zk (K,) int64,
    z_dn list of int64 arrays, docs/freqs lists of lists, labs (D,K) float64."""

    def __init__(self, docs, freqs, labs, V, alpha, beta, z_dn, phantom=False):
        self.docs = [list(map(int, d)) for d in docs]
        self.freqs = [list(map(int, f)) for f in freqs]
        self.labs = np.asarray(labs, dtype=np.float64)
        self.D, self.K = self.labs.shape
        self.V = int(V)
        self.alpha = alpha
        self.beta = beta
        self.z_dn = [np.array(z, dtype=np.int64) for z in z_dn]
        self.n_zk = np.zeros(self.K, dtype=np.int64)
        self.n_d_k = np.zeros((self.D, self.K), dtype=np.int64)
        self.n_k_v = np.zeros((self.K, self.V), dtype=np.int64)
        for d in range(self.D):
            for v, z, f in zip(self.docs[d], self.z_dn[d], self.freqs[d]):
                self.n_zk[z] += f
                self.n_d_k[d, z] += f
                if phantom:
                    # CascadeLDA.py:382-385 iterates the (id, freq) tuples, so n_k_v[z, (id, freq)] += f
                    # bumps column id AND column freq (once if they coincide: fancy-index +=).
                    self.n_k_v[z, (v, f)] += f
                else:
                    self.n_k_v[z, v] += f

    def copy(self):
        s = object.__new__(State)
        s.__dict__.update(self.__dict__)
        s.z_dn = [z.copy() for z in self.z_dn]
        s.n_zk = self.n_zk.copy()
        s.n_d_k = self.n_d_k.copy()
        s.n_k_v = self.n_k_v.copy()
        return s

    def flat_z(self):
        return np.concatenate(self.z_dn) if self.D else np.zeros(0, np.int64)


def init_z_choice(labs, doc_lens, rng=np.random):
    """LabeledLDA.py:85-88: per doc np.random.choice(K, size=len(doc), p=lab/lab.sum())."""
    out = []
    K = labs.shape[1]
    for lab, ld in zip(labs, doc_lens):
        out.append(rng.choice(K, size=ld, p=lab / lab.sum()))
    return out


# ----------------------------------------------------------------------------------------------
# The per-site body (LabeledLDA.py:109-125) -- shared by all modes
# ----------------------------------------------------------------------------------------------
def _site(st, doc_n_d_k, lab, zarr, n, v, f, draw):
    z = zarr[n]
    st.n_k_v[z, v] -= f
    doc_n_d_k[z] -= f
    st.n_zk[z] -= f

    a = doc_n_d_k + st.alpha
    num_b = st.n_k_v[:, v] + st.beta
    den_b = st.n_zk + st.V * st.beta

    prob = lab * a * (num_b / den_b)
    prob /= np.sum(prob)
    z_new = draw(1, prob).argmax()

    zarr[n] = z_new
    st.n_k_v[z_new, v] += f
    doc_n_d_k[z_new] += f
    st.n_zk[z_new] += f


def sweep_sequential(st, draw=None):
    """O1 (draw=None -> np.random.multinomial) / O2 (draw=KeyedDraw with .sweep preset)."""
    native = draw is None
    if native:
        draw = np.random.multinomial
    for d in range(st.D):
        if not native:
            draw.doc = d + getattr(draw, "doc_base", 0)
            draw.site = 0
        row = st.n_d_k[d]
        for n, (v, f) in enumerate(zip(st.docs[d], st.freqs[d])):
            _site(st, row, st.labs[d], st.z_dn[d], n, v, f, draw)


def sweep_snapshot(st, draw, order=None, doc_base=0):
    """O3: per-document snapshot semantics with a keyed draw.  ``order`` = any permutation of docs."""
    snap_kv = st.n_k_v.copy()
    snap_zk = st.n_zk.copy()
    delta_kv = np.zeros_like(snap_kv)
    delta_zk = np.zeros_like(snap_zk)
    order = range(st.D) if order is None else order
    for d in order:
        draw.doc = d + doc_base
        draw.site = 0
        row = st.n_d_k[d]
        ids = st.docs[d]
        for n, (v, f) in enumerate(zip(ids, st.freqs[d])):
            _site(st, row, st.labs[d], st.z_dn[d], n, v, f, draw)
        delta_kv[:, ids] += st.n_k_v[:, ids] - snap_kv[:, ids]
        delta_zk += st.n_zk - snap_zk
        st.n_k_v[:, ids] = snap_kv[:, ids]
        st.n_zk[:] = snap_zk
    st.n_k_v += delta_kv
    st.n_zk += delta_zk


# ----------------------------------------------------------------------------------------------
# Read-outs (reference a7)
# ----------------------------------------------------------------------------------------------
def get_phi(st):
    num = st.n_k_v + st.beta
    den = st.n_zk[:, np.newaxis] + st.V * st.beta
    return num / den


def get_theta(st):
    num = st.n_d_k + st.labs * st.alpha
    den = num.sum(axis=1)[:, np.newaxis]
    return num / den


def perplexity(st):
    """LabeledLDA.py:256-265 -- unweighted by frequency, normalised by the number of sites."""
    phis = get_phi(st)
    thetas = get_theta(st)
    log_per = 0
    l = 0
    for doc, th in zip(st.docs, thetas):
        for w in doc:
            log_per -= np.log(np.inner(phis[:, w], th))
        l += len(doc)
    return np.exp(log_per / l)


def get_ph(st):
    """CascadeLDA.py:394-395 (no smoothing; phantom columns included)."""
    return st.n_k_v / st.n_k_v.sum(axis=1, keepdims=True)


# ----------------------------------------------------------------------------------------------
# Flat (CSR) views used by the C restatement and by the tests that feed the device
# ----------------------------------------------------------------------------------------------
def to_flat(st):
    """-> dict(doc_off int64[D+1], word int32[S], freq int32[S], z int64[S], labs uint8[D,K])."""
    lens = np.array([len(d) for d in st.docs], dtype=np.int64)
    doc_off = np.zeros(st.D + 1, dtype=np.int64)
    np.cumsum(lens, out=doc_off[1:])
    word = np.array([v for d in st.docs for v in d], dtype=np.int32)
    freq = np.array([f for d in st.freqs for f in d], dtype=np.int32)
    return dict(doc_off=doc_off, word=word, freq=freq, z=st.flat_z().astype(np.int64),
                labs=(st.labs != 0).astype(np.uint8))


def digest(n_k_v, n_d_k, n_zk, z_flat):
    """SHA-256 over the four integer arrays as little-endian int64 (reference dtypes)."""
    h = hashlib.sha256()
    for a in (n_k_v, n_d_k, n_zk, z_flat):
        h.update(np.ascontiguousarray(a, dtype="<i8").tobytes())
    return h.hexdigest()


def state_digest(st):
    return digest(st.n_k_v, st.n_d_k, st.n_zk, st.flat_z())


# ----------------------------------------------------------------------------------------------
# Test-time fold-in sampler (reference LabeledLDA.py:155-212)  --  restated
# ----------------------------------------------------------------------------------------------
SWEEP_INIT = 0xFFFFFFFF      # RNG "sweep" word used for the prep4test draws


def prep4test(ph_hat, ids, freqs, draw):
    """LabeledLDA.prep4test (LabeledLDA.py:155-177) for one document given as word ids / frequencies.
    ``draw`` has numpy.random.multinomial's call shape.  Returns (z list, n_dk)."""
    K = ph_hat.shape[0]
    z_dn = []
    n_dk = np.zeros(K, dtype=int)
    probs = ph_hat[:, list(ids)]
    with np.errstate(divide="raise", invalid="raise"):
        try:
            probs /= probs.sum(axis=0)
        except FloatingPointError:
            probs = 1 / K * np.ones_like(probs)
    for n, f in enumerate(freqs):
        prob = probs[:, n]
        while prob.sum() > 1:
            prob /= 1.0000000005
        new_z = draw(1, prob).argmax()
        z_dn.append(new_z)
        n_dk[new_z] += f
    return z_dn, n_dk


def run_test(ph_hat, alpha, docs, freqs, it, thinning, draw_for):
    """LabeledLDA.run_test (LabeledLDA.py:179-212).  ``docs``/``freqs``: lists of id / frequency lists;
    ``draw_for(d, sweep)`` returns the draw callable for document d and sweep (SWEEP_INIT or i)."""
    K = ph_hat.shape[0]
    th_hat = np.zeros((len(docs), K), dtype=float)
    for d, (doc, fr) in enumerate(zip(docs, freqs)):
        z_dn, n_dk = prep4test(ph_hat, doc, fr, draw_for(d, SWEEP_INIT))
        avg_state = None
        for i in range(it):
            draw = draw_for(d, i)
            for n, (v, f, z) in enumerate(zip(doc, fr, z_dn)):
                n_dk[z] -= f
                num_a = n_dk + alpha
                b = ph_hat[:, v]
                prob = num_a * b
                prob /= prob.sum()
                while prob.sum() > 1:
                    prob /= 1.0000005
                new_z = draw(1, prob).argmax()
                z_dn[n] = new_z
                n_dk[new_z] += f
            s = (i + 1) / thinning
            s2 = int(s)
            if s == s2:
                this_state = n_dk / n_dk.sum()
                if s2 == 1:
                    avg_state = this_state
                else:
                    old = (s2 - 1) / s2 * avg_state
                    new = (1 / s2) * this_state
                    avg_state = old + new
                th_hat[d, :] = avg_state
    return th_hat


def keyed_draw_for(seed, stream=0, doc_base=0):
    """draw_for factory: one KeyedDraw per (document, sweep), sites counted from 0."""
    def draw_for(d, sweep):
        k = KeyedDraw(seed, stream)
        k.sweep, k.doc, k.site = sweep, d + doc_base, 0
        return k
    return draw_for


# ----------------------------------------------------------------------------------------------
# CascadeLDA test time (reference CascadeLDA.py:186-247, 299-344)  --  restated
# ----------------------------------------------------------------------------------------------
def cascade_prep4test(ph, beta, ids, freqs, draw):
    ld = len(ids)
    n_dk = np.zeros(ph.shape[0], dtype=int)
    z_dn = []
    probs = ph[:, list(ids)]
    probs += beta
    probs /= probs.sum(axis=0)
    probs[0, :] = 1 / ld
    for n, freq in enumerate(freqs):
        prob = probs[:, n]
        while prob.sum() > 1:
            prob /= 1.0000005
        new_z = draw(1, prob).argmax()
        z_dn.append(new_z)
        n_dk[new_z] += freq
    return z_dn, n_dk


def cascade_test(ph, alpha, beta, ids, freqs, it, thinning, draw_for_sweep):
    """CascadeLDA.cascade_test (CascadeLDA.py:210-247) on the label subset with loadings ph."""
    z_dn, n_dk = cascade_prep4test(ph, beta, ids, freqs, draw_for_sweep(SWEEP_INIT))
    avg_state = np.zeros(ph.shape[0], dtype=float)
    for i in range(it):
        draw = draw_for_sweep(i)
        for n, (v, f, z) in enumerate(zip(ids, freqs, z_dn)):
            n_dk[z] -= f
            num_a = n_dk + alpha
            b = ph[:, v]
            prob = num_a * b
            try:
                with np.errstate(invalid="raise"):
                    prob /= prob.sum()
            except FloatingPointError:
                prob = num_a * (b + beta)
                prob /= prob.sum()
            while prob.sum() > 1:
                prob /= 1.000005
            new_z = draw(1, prob).argmax()
            z_dn[n] = new_z
            n_dk[new_z] += f
        s = (i + 1) / thinning
        s2 = int(s)
        if s == s2:
            this_state = n_dk / n_dk.sum()
            if s2 == 1:
                avg_state = this_state
            else:
                avg_state = (s2 - 1) / s2 * avg_state + (1 / s2) * this_state
    return avg_state


def cascade_run_test(ph, alpha, beta, docs, freqs, it, thinning, draw_for):
    """CascadeLDA.run_test (CascadeLDA.py:299-344), flat test over the rows of ph."""
    th_hat = np.zeros((len(docs), ph.shape[0]), dtype=float)
    for d, (ids, fr) in enumerate(zip(docs, freqs)):
        z_dn, n_zk = cascade_prep4test(ph, beta, ids, fr, draw_for(d, SWEEP_INIT))
        th = None
        for i in range(it):
            draw = draw_for(d, i)
            for n, (v, f) in enumerate(zip(ids, fr)):
                z = z_dn[n]
                n_zk[z] -= f
                num_a = n_zk + alpha
                b = ph[:, v]
                prob = num_a * b
                prob /= prob.sum()
                while prob.sum() > 1:
                    prob /= 1.000005
                new_z = draw(1, prob).argmax()
                z_dn[n] = new_z
                n_zk[new_z] += f
            s = (i + 1) / thinning
            if s == int(s):
                cur_th = n_zk / n_zk.sum()
                if s > 1:
                    m = (s - 1) / s
                    th = m * th + (1 - m) * cur_th
                else:
                    th = cur_th
        th_hat[d, :] = th
    return th_hat
