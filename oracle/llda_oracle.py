"""CPU oracle for the collapsed-Gibbs sweep of LabeledLDA / SubLDA  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``lda_thesis_amd``) never imports anything under ``oracle/`` and raises if the
HIP extension is missing.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the unmodified reference from
``/root/reference`` (gensim stubbed, see ``oracle/refshim.py``), runs its own
``training_iteration`` in the three modes below and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this file against those vectors bit for bit.

Reference lines restated here (all in /root/reference):
  * sweep body           LabeledLDA.py:101-125  ==  CascadeLDA.py:397-421 (SubLDA)
  * count initialisation LabeledLDA.py:80-92, CascadeLDA.py:373-385 (with the phantom-column quirk)
  * read-outs            LabeledLDA.py:231-239 (get_phi/get_theta), :256-265 (perplexity),
                         CascadeLDA.py:394-395 (get_ph)

Three execution modes of the same per-site body:
  O1  sequential, numpy's own legacy stream (np.random.multinomial) -- the reference verbatim.
  O2  sequential, keyed draw  (``draw_keyed`` fed by a Philox uniform keyed on sweep/doc/site).
  O3  per-document snapshot, keyed draw: every document reads the sweep-start n_k_v / n_zk plus its
      own changes; integer deltas are summed afterwards.  This is the semantics the HIP kernel
      implements; it is independent of document order and of the number of GPUs.
"""
import numpy as np

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
    for r in range(10):
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

    counter = (site, doc, stream, sweep), key = (seed & 0xffffffff, seed >> 32);
    u = ((r0 >> 5) * 2**26 + (r1 >> 6)) / 2**53   (all operations exact in float64).
    """
    seed = int(seed)
    r0, r1, _, _ = philox4x32_10(site, doc, stream, sweep, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    a = (r0 >> np.uint64(5)).astype(np.float64)
    b = (r1 >> np.uint64(6)).astype(np.float64)
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)


# ----------------------------------------------------------------------------------------------
# The keyed categorical draw.  Its association order is the one the wavefront uses.
# ----------------------------------------------------------------------------------------------
def elems_per_lane(K):
    """E = topics held by each of the 64 lanes: the smallest power of two with 64*E >= K."""
    if K < 1 or K > 1024:
        raise ValueError("K must be in 1..1024")
    e = 1
    while 64 * e < K:
        e *= 2
    return e


def wave_scan(t):
    """Inclusive scan of 64 float64 values in the order of the CDNA DPP scan:
    row_shr:1,2,4,8 inside each 16-lane row, then row_bcast:15 (rows 1,3), then row_bcast:31 (rows 2,3)."""
    x = np.array(t, dtype=np.float64)
    lane = np.arange(64)
    for sh in (1, 2, 4, 8):
        y = x.copy()
        m = (lane % 16) >= sh
        y[m] = x[lane[m] - sh] + x[m]
        x = y
    x[16:32] = x[15] + x[16:32]
    x[48:64] = x[47] + x[48:64]
    x[32:64] = x[31] + x[32:64]
    return x


def draw_keyed(prob, u):
    """Topic chosen for the normalised probability vector ``prob`` (length K) and uniform ``u``.

    lane l holds topics l*E .. l*E+E-1.  q = sequential inclusive prefix inside the lane,
    X = wave_scan(lane totals), c[k] = X[l-1] + q (X[-1] := 0), t = u * X[63];
    z = the first k with prob[k] > 0 and c[k] > t, or the last k with prob[k] > 0 if there is none.
    """
    prob = np.asarray(prob, dtype=np.float64)
    K = prob.shape[0]
    E = elems_per_lane(K)
    p = np.zeros(64 * E, dtype=np.float64)
    p[:K] = prob
    p = p.reshape(64, E)
    q = p.copy()
    for s in range(1, E):
        q[:, s] = q[:, s - 1] + p[:, s]
    X = wave_scan(q[:, E - 1])
    O = np.concatenate(([0.0], X[:-1]))
    c = O[:, None] + q
    t = u * X[63]
    pos = p > 0.0
    flag = (pos & (c > t)).ravel()
    if flag.any():
        return int(np.argmax(flag))
    pos = pos.ravel()
    if not pos.any():
        raise FloatingPointError("draw_keyed: no positive probability")
    return int(pos.shape[0] - 1 - np.argmax(pos[::-1]))


class KeyedDraw(object):
    """Callable with numpy.random.multinomial's call shape: draw(1, prob) -> one-hot int array.
    Injected into the reference as ``LabeledLDA.multinom_draw`` by gen_golden.py; the driver sets
    ``.sweep``, ``.doc`` (global id) and ``.site`` before each call chain (site auto-increments)."""

    def __init__(self, seed, stream=0):
        self.seed = seed
        self.stream = stream
        self.sweep = 0
        self.doc = 0
        self.site = 0

    def __call__(self, n, prob):
        u = float(keyed_uniform(self.seed, self.sweep, self.stream, self.doc, self.site))
        self.site += 1
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
# Conversions between reference-shaped state and the flat device layout
# ----------------------------------------------------------------------------------------------
def to_device_layout(st):
    """-> dict of flat arrays in the layout include/llda_hip.h documents."""
    K, V, D = st.K, st.V, st.D
    E = elems_per_lane(K)
    KP = 64 * E
    lens = np.array([len(d) for d in st.docs], dtype=np.int64)
    doc_off = np.zeros(D + 1, dtype=np.int64)
    np.cumsum(lens, out=doc_off[1:])
    word = np.array([v for d in st.docs for v in d], dtype=np.int32)
    freq = np.array([f for d in st.freqs for f in d], dtype=np.int32)
    z = st.flat_z().astype(np.int32)
    n_dk = np.zeros((D, KP), dtype=np.int32)
    n_dk[:, :K] = st.n_d_k
    n_kw = np.zeros((V, KP), dtype=np.int32)
    n_kw[:, :K] = st.n_k_v.T
    n_k = np.zeros(KP, dtype=np.int32)
    n_k[:K] = st.n_zk
    lab_bits = pack_label_bits(st.labs, KP)
    return dict(doc_off=doc_off, word=word, freq=freq, z=z, n_dk=n_dk, n_kw=n_kw, n_k=n_k,
                lab_bits=lab_bits, K=K, V=V, D=D, KP=KP, E=E)


def pack_label_bits(labs, KP):
    """(D,K) 0/1 -> (D, KP/32) uint32, bit (k % 32) of word k // 32 set iff labs[d,k] != 0."""
    labs = np.asarray(labs)
    D, K = labs.shape
    bits = np.zeros((D, KP), dtype=np.uint8)
    bits[:, :K] = labs != 0
    bits = bits.reshape(D, KP // 32, 32).astype(np.uint32)
    return (bits << np.arange(32, dtype=np.uint32)).sum(axis=2, dtype=np.uint64).astype(np.uint32)


def digest(n_k_v, n_d_k, n_zk, z_flat):
    """SHA-256 over the four integer arrays as little-endian int64 (reference dtypes)."""
    import hashlib
    h = hashlib.sha256()
    for a in (n_k_v, n_d_k, n_zk, z_flat):
        h.update(np.ascontiguousarray(a, dtype="<i8").tobytes())
    return h.hexdigest()
