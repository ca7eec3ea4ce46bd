"""CascadeLDA's ensemble of per-node Labeled-LDA sub-problems trained TOGETHER on one device.

The reference trains the 122 sub-problems of the abstracts corpus one after another
(/root/reference/CascadeLDA.py:135-184, each a SubLDA, CascadeLDA.py:347-434).  They share nothing but the read-only
corpus, and most are tiny (6 .. 4171 documents, 2 .. 20 topics), so here all of them live in ONE set of device
buffers and every sweep of the whole ensemble is at most four launches of ``llda_sweep_batch``
(include/llda_gibbs.h) plus one ``llda_apply_delta`` (a site the kernel's margin cannot decide goes through the exact
fp64 pipeline inside the kernel, so the result is always the reference's):

    document instance i = (sub-problem p, member document d)      I = sum_p D_p instances, visiting order
    inst_off (I+1), word / freq / z (S_tot)                       CSR over instance sites (z = device positions)
    counts / delta                                                per problem [n_kw (V x KP_p) | n_k (KP_p)], fused
    n_dk                                                          per instance a KP_p row
    live_off / live_pos                                           allowed positions of every instance, in draw order

Host side, everything the reference does per sub-problem in python loops is done once, vectorised:
  * the initial assignments -- the reference draws ``np.random.choice(K, size=len(doc), p=lab/lab.sum())`` per
    document (CascadeLDA.py:373-381); numpy's legacy ``choice`` is ``cdf = p.cumsum(); cdf /= cdf[-1];
    cdf.searchsorted(random_sample(n), 'right')``, so ONE ``random_sample`` of all sites of all sub-problems in
    visiting order consumes the global stream identically, and the searchsorted becomes a table lookup;
  * the count initialisation incl. the phantom columns (CascadeLDA.py:382-385: ``n_k_v[z, (id, f)] += f`` bumps
    column ``f`` as well as column ``id``) -- a few ``index_add_`` calls over all instance sites;
  * ``get_ph`` of every problem (CascadeLDA.py:394-395) and the scatter of its rows into ``ph`` (CascadeLDA.py:143-184).
The RNG key of a site is (seed; sweep, stream = index of the sub-problem in visiting order, document index inside
the sub-problem, site) -- the key SubLDA uses when the problems are trained one by one, so both paths are
bit-identical (tests/test_gpu_dropin.py).
"""
import numpy as np
import torch

from . import _native
from .layout import group_layout

MAX_BATCH_K = 128            # llda_sweep_batch: every problem is one numpy pairwise leaf (8 lanes)
MAX_BATCH_ALLOWED = 64       # ... and a document has one lane per allowed topic


def choice_cdf_table(a_max):
    """tab[A-1, j] = the normalised cdf numpy's legacy choice builds for A equally likely categories
    (p = 1.0 / A each; zeros in between do not change a cumulative sum), +inf beyond A."""
    tab = np.full((a_max, a_max), np.inf)
    for A in range(1, a_max + 1):
        p = np.ones(A) / float(A)            # lab / lab.sum()
        c = p.cumsum()
        c /= c[-1]
        tab[A - 1, :A] = c
    return tab


def draw_initial_topics(allowed, n_allowed, inst_of_site, uniforms):
    """np.random.choice(K, size=n, p=lab/lab.sum()) for every instance at once: ``allowed`` (I, A_max) local topic ids
    ascending (padded), ``n_allowed`` (I,), one uniform per site.  -> local topic per site."""
    tab = choice_cdf_table(int(n_allowed.max()))
    a_site = n_allowed[inst_of_site]
    k = np.zeros(uniforms.shape[0], dtype=np.int64)
    for A in np.unique(a_site):                                    # one searchsorted per label-set size
        m = a_site == A
        k[m] = np.searchsorted(tab[A - 1, :A], uniforms[m], side="right")
    return allowed[inst_of_site, k]


def draw_initial_topics_device(allowed, n_allowed, inst_len, uniforms, device):
    """draw_initial_topics on the device: ``allowed`` (I, A_max) / ``n_allowed`` (I,) / ``inst_len`` (I,) host arrays, the
    uniforms of numpy's stream (host, one per site) -> local topic of every site as an int64 device tensor.
    numpy's ``searchsorted(cdf, u, 'right')`` is k = #{j : cdf[j] <= u}; the cdf of A equal weights is within a few ulps of
    (j + 1) / A, so floor(u * A) is at most one off and two exact comparisons against the SAME table settle it."""
    a_max = int(n_allowed.max())
    tab = torch.from_numpy(choice_cdf_table(a_max)).to(device).reshape(-1)
    al = torch.from_numpy(np.ascontiguousarray(allowed)).to(device)
    na = torch.from_numpy(np.ascontiguousarray(n_allowed)).to(device)
    inst = torch.repeat_interleave(torch.arange(al.shape[0], device=device), torch.from_numpy(np.ascontiguousarray(inst_len)).to(device))
    u = torch.from_numpy(uniforms).to(device)
    A = na[inst]
    row = (A - 1) * a_max
    k = torch.minimum((u * A.to(torch.float64)).to(torch.int64), A - 1)
    for _ in range(3):
        too_low = tab[row + k] <= u                                   # cdf[k] <= u: the answer is beyond k
        too_high = (k > 0) & (tab[row + torch.clamp(k - 1, min=0)] > u)
        k = k + too_low.to(torch.int64) - too_high.to(torch.int64)
    ok = (tab[row + k] > u) & ((k == 0) | (tab[row + torch.clamp(k - 1, min=0)] <= u))
    if not bool(ok.all()):
        raise AssertionError("draw_initial_topics_device: a draw did not settle")
    return al[inst, k]


class Ensemble(object):
    """Device state of a set of sub-problems (``plans``: see CascadeLDA.plan_subproblems) and the sweep driver."""

    def __init__(self, plans, z_local, doc_off, word, freq, V, alpha, beta, seed, device=None, streams=None):
        """streams[i] = RNG stream of plans[i] (default i).  plans[i]: dict(K, docs (member document ids), allowed (D_p, A_max) local topics ascending, padded
        with -1, n_allowed (D_p,)); z_local[i]: initial local topic of every site of plan i (instance order) -- or ONE int64 device
        tensor holding those of all plans, in that order, or a function of the device that returns it."""
        _native.lib()
        _native.require_device()
        self.device = dev = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.V, self.alpha, self.beta, self.seed = int(V), float(alpha), float(beta), int(seed)
        self.plans = plans
        self.sweeps_done = 0
        self.debug_margin = 0
        P = len(plans)
        lens_all = np.diff(doc_off)
        layouts = {}
        for pl in plans:
            if pl["K"] not in layouts:
                layouts[pl["K"]] = group_layout(pl["K"])
        if any(pl["K"] > MAX_BATCH_K for pl in plans):
            raise ValueError("the batched ensemble handles sub-problems of at most %d topics" % MAX_BATCH_K)
        kp = np.array([layouts[pl["K"]].KP for pl in plans], dtype=np.int64)
        n_docs = np.array([len(pl["docs"]) for pl in plans], dtype=np.int64)
        # fused [n_kw | n_k] per problem
        blk = (self.V + 1) * kp
        kw_off = np.concatenate(([0], np.cumsum(blk)))[:-1]
        nk_off = kw_off + self.V * kp
        self.total = int(blk.sum())
        inst_prob = np.repeat(np.arange(P), n_docs)
        inst_doc = np.concatenate([np.arange(n) for n in n_docs]) if P else np.zeros(0, np.int64)
        inst_gdoc = np.concatenate([pl["docs"] for pl in plans]) if P else np.zeros(0, np.int64)
        inst_len = lens_all[inst_gdoc]
        inst_off = np.concatenate(([0], np.cumsum(inst_len)))
        self.I, self.S = int(inst_prob.shape[0]), int(inst_off[-1])
        ndk_off = np.concatenate(([0], np.cumsum(kp[inst_prob])))
        self.ndk_total = int(ndk_off[-1])
        # sites of every instance: gather from the corpus CSR
        site_src = np.repeat(doc_off[inst_gdoc] - inst_off[:-1], inst_len) + np.arange(self.S)
        inst_of_site = np.repeat(np.arange(self.I), inst_len)
        w_s, f_s = word[site_src], freq[site_src]
        if f_s.size and int(f_s.max()) >= self.V:
            raise IndexError("index %d is out of bounds for axis 1 with size %d" % (int(f_s.max()), self.V))
        if f_s.size and (int(f_s.min()) < 0 or int(f_s.max()) >= (1 << 23) or int(f_s.astype(np.int64).sum()) >= (1 << 31)):
            raise ValueError("word frequencies must be in [0, 2^23) and the ensemble must hold fewer than 2^31 tokens")
        # local topic -> device position, per problem layout
        a_max = max(pl["allowed"].shape[1] for pl in plans)
        pos_tab = np.zeros((P, max(pl["K"] for pl in plans)), dtype=np.int64)
        rank_tab = np.zeros_like(pos_tab)                        # place of the topic in the draw order ((lane, slot))
        for i, pl in enumerate(plans):
            pos_tab[i, :pl["K"]] = layouts[pl["K"]].topic_pos
            rank_tab[i, :pl["K"]] = layouts[pl["K"]].lm_topic_pos
        if callable(z_local):
            z_local = z_local(dev)
        z_on_device = isinstance(z_local, torch.Tensor)
        if not z_on_device:
            z_loc = np.concatenate(z_local) if P else np.zeros(0, np.int64)
            z_pos = pos_tab[inst_prob[inst_of_site], z_loc]
        # allowed positions of every instance, in draw order
        allowed = np.full((self.I, a_max), -1, dtype=np.int64)
        row = 0
        for pl in plans:
            n, a = pl["allowed"].shape
            allowed[row:row + n, :a] = pl["allowed"]
            row += n
        valid = allowed >= 0
        big = np.iinfo(np.int64).max
        akey = np.where(valid, rank_tab[inst_prob[:, None], np.maximum(allowed, 0)], big)
        order = np.argsort(akey, axis=1, kind="stable")
        apos = np.take_along_axis(np.where(valid, pos_tab[inst_prob[:, None], np.maximum(allowed, 0)], big), order, axis=1)
        n_allowed = valid.sum(axis=1)
        live_off = np.concatenate(([0], np.cumsum(n_allowed)))
        live_pos = apos[apos != big]
        self.n_allowed_max = int(n_allowed.max()) if self.I else 0
        if self.n_allowed_max > MAX_BATCH_ALLOWED:
            raise ValueError("a document allows %d topics; the batched ensemble handles at most %d (CascadeLDA.go_down_tree "
                             "takes the one-by-one path for such corpora)" % (self.n_allowed_max, MAX_BATCH_ALLOWED))

        def up(a, dtype):
            return torch.from_numpy(np.ascontiguousarray(a)).to(device=dev, dtype=dtype)

        self.inst_off, self.word, self.freq = up(inst_off, torch.int64), up(w_s, torch.int32), up(f_s, torch.int32)
        self.inst_prob, self.inst_doc = up(inst_prob, torch.int32), up(inst_doc, torch.int32)
        i_s = up(inst_of_site, torch.int64)
        if z_on_device:
            self.z = up(pos_tab, torch.int64)[self.inst_prob.to(torch.int64)[i_s], z_local.to(dev)].to(torch.int32)
        else:
            self.z = up(z_pos, torch.int32)
        self.live_off, self.live_pos = up(live_off, torch.int64), up(live_pos, torch.int32)
        self.ndk_off = up(ndk_off[:-1], torch.int64)
        self.kw_off, self.nk_off, self.kp = up(kw_off, torch.int64), up(nk_off, torch.int64), up(kp, torch.int32)
        self.prob_stream = up(np.arange(P) if streams is None else np.asarray(streams), torch.int32)
        self.prob_k = up(np.array([pl["K"] for pl in plans]), torch.int32)
        self._kp_h, self._kw_off_h, self._nk_off_h, self._ndk_off_h = kp, kw_off, nk_off, ndk_off
        self._inst_off_h, self._n_docs_h, self._layouts = inst_off, n_docs, layouts
        self.status = torch.zeros((4,), dtype=torch.int32, device=dev)
        # instances grouped by the lanes they need (8 / 16 / 32 / 64), longest first inside a group
        lanes = np.where(n_allowed <= 8, 8, np.where(n_allowed <= 16, 16, np.where(n_allowed <= 32, 32, 64)))
        self.orders = []
        for g in (8, 16, 32, 64):
            ids = np.flatnonzero(lanes == g)
            if ids.size:
                ids = ids[np.argsort(-inst_len[ids], kind="stable")]
                self.orders.append((g, up(ids, torch.int32)))
        # counts from the assignments (CascadeLDA.py:382-385), phantom columns included
        self.counts = torch.zeros((self.total,), dtype=torch.int32, device=dev)
        self.delta = torch.zeros((self.total,), dtype=torch.int32, device=dev)
        self.n_dk = torch.zeros((max(self.ndk_total, 1),), dtype=torch.int32, device=dev)
        if self.S:
            p_s = self.inst_prob.to(torch.int64)[i_s]
            kp_s = self.kp.to(torch.int64)[p_s]
            z64, w64, f64 = self.z.to(torch.int64), self.word.to(torch.int64), self.freq.to(torch.int64)
            self.counts.index_add_(0, self.kw_off[p_s] + w64 * kp_s + z64, self.freq)
            ghost = f64 != w64                                     # n_k_v[z, (id, f)] += f touches column f too
            self.counts.index_add_(0, (self.kw_off[p_s] + f64 * kp_s + z64)[ghost], self.freq[ghost])
            self.counts.index_add_(0, self.nk_off[p_s] + z64, self.freq)
            self.n_dk.index_add_(0, self.ndk_off[i_s] + z64, self.freq)

    # ------------------------------------------------------------------ the hot path
    def sweep(self):
        """one Gibbs sweep of every sub-problem (SubLDA.training_iteration, CascadeLDA.py:397-421)."""
        for lanes, order in self.orders:
            _native.sweep_batch(inst_off=self.inst_off, order=order, word=self.word, freq=self.freq, z=self.z,
                                inst_prob=self.inst_prob, inst_doc=self.inst_doc, live_off=self.live_off,
                                live_pos=self.live_pos, ndk_off=self.ndk_off, n_dk=self.n_dk, kw_off=self.kw_off,
                                nk_off=self.nk_off, kp=self.kp, prob_stream=self.prob_stream, k=self.prob_k, counts=self.counts, delta=self.delta,
                                status=self.status, V=self.V, lanes=lanes, alpha=self.alpha, beta=self.beta,
                                seed=self.seed, sweep=self.sweeps_done, debug_margin=self.debug_margin)
        _native.apply_delta(self.counts, self.delta)
        self.sweeps_done += 1

    def check_status(self):
        """Raise like the reference would (numpy's multinomial rejects a NaN pvals vector); synchronises."""
        if int(self.status[0].item()) & 1:
            raise ValueError("a site had no topic with positive probability (pvals would be NaN)")

    # ------------------------------------------------------------------ read-outs
    def ph_rows(self):
        """get_ph of every problem (CascadeLDA.py:394-395: n_k_v / n_k_v.sum(axis=1), no smoothing, phantom columns
        included) as a list of (K_p, V) float64 device tensors in the reference's topic order."""
        out = []
        for i, pl in enumerate(self.plans):
            kp, K = int(self._kp_h[i]), pl["K"]
            n_kw = self.counts[int(self._kw_off_h[i]):int(self._kw_off_h[i]) + self.V * kp].view(self.V, kp)
            tp = torch.from_numpy(self._layouts[K].topic_pos.astype(np.int64)).to(self.device)
            rows = n_kw[:, tp].t()                                  # (K, V) int32
            den = rows.sum(dim=1, dtype=torch.int64).to(torch.float64)
            out.append(rows.to(torch.float64) / den[:, None])
        return out

    def scatter_ph(self, ph_dev, label_rows):
        """ph_dev[label_rows[i][j]] = get_ph()[j] of problem i for every j with label_rows[i][j] >= 0 -- grouped by
        row length so that the whole ensemble is a handful of tensor operations."""
        by_kp = {}
        for i in range(len(self.plans)):
            by_kp.setdefault(int(self._kp_h[i]), []).append(i)
        for kp, probs in by_kp.items():
            # (n, V, kp) view of the n_kw blocks of these problems (each block is (V + 1) * kp long)
            starts = torch.from_numpy(self._kw_off_h[probs]).to(self.device)
            idx = starts[:, None] + torch.arange(self.V * kp, device=self.device)[None, :]
            blocks = self.counts[idx].view(len(probs), self.V, kp)
            den = blocks.sum(dim=1, dtype=torch.int64).to(torch.float64)            # (n, kp) column sums
            src_p, src_pos, dst = [], [], []
            for j, i in enumerate(probs):
                K = self.plans[i]["K"]
                tp = self._layouts[K].topic_pos
                for t in range(K):
                    r = label_rows[i][t]
                    if r >= 0:
                        src_p.append(j); src_pos.append(int(tp[t])); dst.append(r)
            if not dst:
                continue
            sp = torch.tensor(src_p, device=self.device)
            so = torch.tensor(src_pos, device=self.device)
            num = blocks[sp, :, so].to(torch.float64)                               # (rows, V)
            ph_dev[torch.tensor(dst, device=self.device)] = num / den[sp, so][:, None]

    # ------------------------------------------------------------------ reference-layout views (tests)
    def problem_state(self, i):
        """(n_k_v (K,V), n_d_k (D,K), n_zk (K,), z flat topic ids) of problem i as int64 numpy arrays."""
        pl, kp, K = self.plans[i], int(self._kp_h[i]), self.plans[i]["K"]
        lay = self._layouts[K]
        tp = torch.from_numpy(lay.topic_pos.astype(np.int64)).to(self.device)
        kw0, nk0 = int(self._kw_off_h[i]), int(self._nk_off_h[i])
        n_k_v = self.counts[kw0:kw0 + self.V * kp].view(self.V, kp)[:, tp].t().contiguous().cpu().numpy().astype(np.int64)
        n_zk = self.counts[nk0:nk0 + kp][tp].cpu().numpy().astype(np.int64)
        i0 = int(self._n_docs_h[:i].sum())
        D = int(self._n_docs_h[i])
        d0 = int(self._ndk_off_h[i0])
        n_d_k = self.n_dk[d0:d0 + D * kp].view(D, kp)[:, tp].cpu().numpy().astype(np.int64)
        s0, s1 = int(self._inst_off_h[i0]), int(self._inst_off_h[i0 + D])
        pt = torch.from_numpy(lay.pos_topic.astype(np.int64)).to(self.device)
        z = pt[self.z[s0:s1].to(torch.int64)].cpu().numpy().astype(np.int64)
        return n_k_v, n_d_k, n_zk, z
