"""GPU parity: the HIP sweep (through the C ABI) against golden vectors produced by the unmodified
reference under per-document snapshot semantics (O3), and against the C oracle on seeded inputs.
Bit-exact for every integer array after every sweep."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from helpers import assert_state_equal, c_state

pytestmark = pytest.mark.gpu

TINY = golden_names("tiny_")


def make_sampler(g, counts=True, **kw):
    from lda_thesis_amd.sampler import GibbsSampler
    c = dict(n_d_k=g["init_n_d_k"], n_k_v=g["init_n_k_v"], n_zk=g["init_n_zk"]) if counts else None
    return GibbsSampler(g["doc_off"], g["word"], g["freq"], g["init_z"], int(g["K"]), int(g["V"]),
                        float(g["alpha"]), float(g["beta"]), labs=g["labs"], counts=c,
                        seed=int(g["seed"]), stream_id=int(g["stream"]) if "stream" in g else 0, **kw)


# debug_margin: 0 = production tiered draw (fp32 decision when |Q - T| > 2^-17 of the total, else fp64
# decision when > 2^-40, else the exact fp64 pipeline); -1 = every site through the exact tier; -2 = no fp32
# tier; 6 = both margins 2^-6, i.e. many sites fall through, mixing all tiers inside one wavefront
@pytest.mark.parametrize("margin", [0, -1, -2, 6])
@pytest.mark.parametrize("name", TINY + ["sublda"])
def test_sweeps_match_reference_o3(name, margin):
    g = load_golden(name)
    s = make_sampler(g)
    s.debug_margin = margin
    for i in range(int(g["sweeps"])):
        s.sweep()
        assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics(), name)
    s.check_status()


@pytest.mark.parametrize("name", ["tiny_k512dense", "tiny_k1024dense"])
def test_headline_kernels_match_reference_o3(name):
    """every label in every document (a dense mask, K == KP) with the commit log: the kernels of the bench's timed line -- at K = 512 four
    documents per wavefront on the image of every row (quad None / True: production margins, margins 2^-6 that mix the tiers inside a
    wavefront, no fp32 tier, everything through the exact tier), the two-document 16-bit-row kernel at four waves per SIMD (quad False)
    and at three (debug_margin -8), int32 rows -- against the reference's own O3 sweeps"""
    g = load_golden(name)
    k512 = int(g["K"]) == 512
    for rows16, margin, quad in ((None, 0, None), (True, 0, True if k512 else None), (True, 6, None), (True, -2, None), (True, -1, None),
                                 (True, 0, False), (True, -8, False), (True, 6, False), (True, -1, False), (False, 0, None)):
        s = make_sampler(g, commit_log=True, rows16=rows16, quad=quad)
        assert s.dense_mask and s.commit_log is not None and (s.n_kw16 is not None) == (rows16 is not False)
        assert s.quad == (k512 and rows16 is not False and quad is not False)
        if rows16 is not False:
            assert 0 < s.max_doc_tokens < 65536
            if not s.quad:
                assert bool(s.row16.all())
        s.debug_margin = margin
        for i in range(int(g["sweeps"])):
            s.sweep()
            assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics(), name)
        s.check_status()
        if s.quad:
            assert bool(s.row16.all()) and s.site_row is None           # (the library's flags: every row of the fixture fits)


@pytest.mark.parametrize("name", ["tiny_k256dense", "tiny_k128dense", "tiny_k100dense", "tiny_k200dense", "tiny_k400dense"])
def test_narrow_quad_kernels_match_reference_o3(name):
    """K = 256 / 128, every label in every document, commit log: llda_sweep_quad_kernel<3> / <2> -- eight / sixteen documents per
    wavefront, 16-byte site records, own count out of the packed row -- against the reference's own O3 sweeps, in every tier mode; and
    the general kernel (quad False) on the same fixture.  K = 100, 200, 250, 400: the same kernels on layouts with positions that hold no
    topic (K < KP: one leaf with a tail, two unequal leaves, four unequal leaves) --
    the exact tier (margin -1: every site) sums in numpy's order for K."""
    g = load_golden(name)
    for margin, quad in ((0, None), (6, True), (-2, True), (-1, True), (0, False)):
        s = make_sampler(g, commit_log=True, quad=quad)
        assert s.dense_mask and s.commit_log is not None and (s.site_rec is not None) == (s.layout.G <= 16)
        assert s.quad == (quad is not False) and (s.n_kw16 is not None) == s.quad
        s.debug_margin = margin
        for i in range(int(g["sweeps"])):
            s.sweep()
            assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics(), name)
        s.check_status()
        if s.quad:
            assert bool(s.row16.all()) and s.site_row is None and 0 < s.max_doc_tokens < 65536


@pytest.mark.parametrize("name", [n for n in TINY if int(n.split("k")[-1].rstrip("dense")) > 1024])
def test_wide_layout_kernels_agree_with_reference(c_oracle, name):
    """wide layouts (more than 8 pairwise leaves): the LDS-only tiered kernel (debug_margin -3; production runs the one that
    keeps the row in registers) and the all-exact kernel that tiny priors select -- the fixtures' priors are not tiny, so
    the latter is checked against the C oracle on the fixture's corpus with alpha = beta = 1e-9."""
    from lda_thesis_amd.sampler import GibbsSampler
    g = load_golden(name)
    # production (0) runs the fp32-tiered kernel; -7: with fp32 factors only in LDS (rare tiers on the scratch row); -6: with
    # fp64 factors in LDS; -5: the fp64 kernel with the row in registers and int16 changes; -4: the same with LDS copies of
    # the counts; -3: the LDS-only kernel
    for margin in (0, -7, -6, -5, -3, -4):
        s = make_sampler(g)
        assert s.layout.wide and 0 < s.max_doc_tokens < 32768 and (s._scratch is not None) == (s.live_off is None)
        s.debug_margin = margin
        for i in range(int(g["sweeps"])):
            s.sweep()
            assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics(), name)
        s.check_status()
    K, V = int(g["K"]), int(g["V"])
    t = GibbsSampler(g["doc_off"], g["word"], g["freq"], g["init_z"], K, V, 1e-9, 1e-9, labs=g["labs"], seed=5)
    cs = c_oracle.CState(g["doc_off"], g["word"], g["freq"], g["init_z"], g["labs"], t.n_d_k(), t.n_k_v(), t.n_zk(), V,
                         1e-9, 1e-9)
    for i in range(2):
        t.sweep()
        cs.sweep(1, 5, i, threads=2)
        np.testing.assert_array_equal(t.z_topics(), cs.z)
        np.testing.assert_array_equal(t.n_k_v(), cs.n_k_v)
    t.check_status()


def test_wide_layout_documents_too_heavy_for_int16_changes(c_oracle):
    """a document of 2^15 tokens or more: the sampler's max_doc_tokens hint keeps the wide kernel on full-width counts."""
    from lda_thesis_amd.sampler import GibbsSampler
    rng = np.random.default_rng(3)
    K, D, V = 1500, 12, 200
    doc_off, word, freq, labs, z = synth(rng, D, V, K, 1, 40, True)
    freq = freq.copy()
    freq[doc_off[3]:doc_off[4]] = 9000                  # document 3: tens of thousands of tokens
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=8)
    assert s.layout.wide and s.max_doc_tokens >= 32768
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(3):
        s.sweep()
        cs.sweep(1, 8, i, threads=2)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
    s.check_status()


@pytest.mark.parametrize("commit", ["atomics", "log_items_of_16", "log_items_of_3"])
@pytest.mark.parametrize("name", TINY + ["sublda"])
def test_commit_paths_agree(name, commit, monkeypatch):
    """n_kw is updated either by int32 atomics from inside the sweep kernels or by folding the word-major commit
    log (llda_commit_log); hot words are cut into several log items whose rows are combined with atomics."""
    from lda_thesis_amd.sampler import GibbsSampler
    g = load_golden(name)
    if commit != "atomics":
        monkeypatch.setattr(GibbsSampler, "LOG_ITEM", int(commit.rsplit("_", 1)[1]))
    s = make_sampler(g, commit_log=commit != "atomics")
    assert (s.commit_log is None) == (commit == "atomics")
    if commit != "atomics":
        if commit.endswith("_3"):
            assert int((s.item_word < 0).sum()) > 0                 # some word is spread over several items
        assert int(s.item_len.sum()) == s.S and int(s.item_len.max()) <= GibbsSampler.LOG_ITEM
    for i in range(int(g["sweeps"])):
        s.sweep()
        assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics(), name)
    s.check_status()


@pytest.mark.parametrize("name", ["tiny_k40", "tiny_k130", "tiny_k392", "tiny_k512", "sublda", "tiny_k512dense", "tiny_k256dense",
                                  "tiny_k128dense", "tiny_k100dense"])          # (the dense ones with the commit log: the quad kernel, 4 / 8 / 16 documents per wavefront)
def test_shard_split_into_several_calls(name, monkeypatch):
    """llda_sweep addresses the sites of one call with 32-bit offsets from its first document, so a shard that
    spans 2^30 sites is walked in several calls over document ranges; here the limit is lowered to 150 sites."""
    from lda_thesis_amd.sampler import GibbsSampler
    g = load_golden(name)
    monkeypatch.setattr(GibbsSampler, "MAX_CALL_SITES", 150)
    for commit in (False, True):
        s = make_sampler(g, commit_log=commit)
        assert s.quad == (commit and name.endswith("dense"))
        assert len(s._calls) > 2 and s._calls[0][0] == 0 and s._calls[-1][1] == s.D
        assert all(a[1] == b[0] for a, b in zip(s._calls, s._calls[1:]))
        for i in range(int(g["sweeps"])):
            s.sweep()
            assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics(), name)
        s.check_status()


def test_call_spanning_too_many_sites_is_refused():
    from lda_thesis_amd import _native
    g = load_golden("tiny_k40")
    s = make_sampler(g)
    kw = dict(doc_off=s.doc_off, doc_order=None, word=s.word, freq=s.freq, z=s.z, lab_mask=s.lab_mask, n_dk=s.n_dk,
              n_kw=s.n_kw, n_kw_delta=s.n_kw_delta, n_k=s.n_k, n_k_delta=s.n_k_delta, status=s.status, D=s.D, V=s.V,
              K=s.K, alpha=s.alpha, beta=s.beta, seed=1, sweep=0)
    with pytest.raises(_native.NativeError):
        _native.sweep(n_sites=1 << 30, **kw)


@pytest.mark.parametrize("name", TINY)
def test_count_init_matches_reference(name):
    g = load_golden(name)
    s = make_sampler(g, counts=False)
    np.testing.assert_array_equal(s.n_k_v(), g["init_n_k_v"])
    np.testing.assert_array_equal(s.n_d_k(), g["init_n_d_k"])
    np.testing.assert_array_equal(s.n_zk(), g["init_n_zk"])


@pytest.mark.parametrize("name", TINY)
def test_label_csr_equals_dense_labs(name):
    g = load_golden(name)
    from lda_thesis_amd.sampler import GibbsSampler
    rows, cols = np.nonzero(g["labs"])
    off = np.zeros(int(g["D"]) + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=int(g["D"])), out=off[1:])
    a = make_sampler(g)
    b = GibbsSampler(g["doc_off"], g["word"], g["freq"], g["init_z"], int(g["K"]), int(g["V"]),
                     float(g["alpha"]), float(g["beta"]), labs=(off, cols), seed=int(g["seed"]))
    assert (a.lab_mask == b.lab_mask).all()


@pytest.mark.parametrize("name", TINY)
def test_perplexity_matches_reference(name):
    """log-likelihood read-out: 1e-5 relative is the bar in BASELINE.json; we hold 1e-9."""
    g = load_golden(name)
    s = make_sampler(g)
    for _ in range(int(g["sweeps"])):
        s.sweep()
    assert abs(s.perplexity() / float(g["o3_perplexity"]) - 1.0) < 1e-9


@pytest.mark.parametrize("docs_per_group", [1, 3])
@pytest.mark.parametrize("sort_docs", [False, True])
def test_schedule_independence(docs_per_group, sort_docs):
    """work distribution (document order, documents per group) must not change the result."""
    g = load_golden("tiny_k40")
    s = make_sampler(g, docs_per_group=docs_per_group, sort_docs=sort_docs)
    for i in range(int(g["sweeps"])):
        s.sweep()
    assert_state_equal(g, "o3_s%d" % int(g["sweeps"]), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics())


def synth(rng, D, V, K, n_lo, n_hi, dense, fmax=3):
    lens = rng.integers(n_lo, n_hi + 1, size=D)
    doc_off = np.zeros(D + 1, dtype=np.int64)
    np.cumsum(lens, out=doc_off[1:])
    word = np.concatenate([np.sort(rng.choice(V, size=n, replace=False)) for n in lens]).astype(np.int32)
    freq = rng.integers(1, fmax + 1, size=int(doc_off[-1])).astype(np.int32)
    labs = np.ones((D, K), dtype=np.uint8)
    if not dense:
        labs = (rng.random((D, K)) < min(1.0, 6.0 / K)).astype(np.uint8)
        labs[:, 0] = 1
    z = np.empty(int(doc_off[-1]), dtype=np.int64)
    for d in range(D):
        allowed = np.nonzero(labs[d])[0]
        z[doc_off[d]:doc_off[d + 1]] = rng.choice(allowed, size=lens[d])
    return doc_off, word, freq, labs, z


@pytest.mark.parametrize("K,dense,D,V", [(7, True, 300, 200), (64, True, 400, 500), (128, True, 300, 1000),
                                         (392, False, 300, 800), (512, True, 96, 2000), (1024, True, 40, 600),
                                         (300, True, 64, 500), (100, False, 1000, 300),
                                         # every (lanes per document, slots per lane, tail) shape of the layout
                                         (3, True, 200, 100), (15, False, 200, 150), (17, True, 150, 200),
                                         (33, True, 150, 200), (60, True, 120, 300), (90, True, 120, 300),
                                         (96, True, 100, 300), (129, True, 100, 300), (190, False, 100, 400),
                                         (256, True, 80, 400), (257, True, 80, 400), (640, True, 50, 500),
                                         (777, True, 40, 500), (900, True, 40, 500), (968, True, 32, 500),
                                         # wide layouts: more than 8 pairwise leaves, 2 .. 8 tiers of 64 lanes (the last
                                         # three need more than 64 KB of LDS per wavefront)
                                         (969, True, 32, 500), (1500, False, 40, 400), (2047, True, 24, 300),
                                         (4096, True, 16, 300), (5000, False, 24, 300), (6000, True, 12, 200),
                                         (7688, True, 12, 200)])
def test_seeded_inputs_vs_c_oracle(c_oracle, K, dense, D, V):
    """larger seeded inputs: HIP == C oracle (snapshot mode) after 3 sweeps, every integer."""
    from lda_thesis_amd.sampler import GibbsSampler
    rng = np.random.default_rng(K * 7 + D)
    doc_off, word, freq, labs, z = synth(rng, D, V, K, 1, 90, dense)
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=99, doc_base=1000)
    s.debug_margin = [0, 6, -1, -2][(K + D) % 4]
    n_dk0, n_kv0, n_k0 = s.n_d_k(), s.n_k_v(), s.n_zk()
    cs = c_oracle.CState(doc_off, word, freq, z, labs, n_dk0, n_kv0, n_k0, V, 0.1, 0.01)
    for i in range(3):
        s.sweep()
        cs.sweep(1, 99, i, doc_base=1000, threads=4)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
    s.check_status()
    # invariants
    tot = int((freq.astype(np.int64)).sum())
    assert s.n_zk().sum() == tot and s.n_k_v().sum() == tot
    assert (s.n_d_k().sum(0) == s.n_zk()).all() and (s.n_k_v().sum(1) == s.n_zk()).all()
    assert (s.n_d_k()[labs == 0] == 0).all() and s.n_d_k().min() >= 0 and s.n_k_v().min() >= 0


@pytest.mark.parametrize("K,dense", [(96, True), (128, True), (192, True), (256, True), (384, True), (512, True), (768, True),
                                     (1024, True), (100, False), (250, False), (392, False), (500, False), (777, False),
                                     (1000, False)])
def test_logged_instantiations_vs_c_oracle(c_oracle, K, dense):
    """the commit-log instantiations of the tiered kernel for every layout class -- 8 / 16 / 32 / 64 lanes per document x
    12 / 16 slots per lane: three or four waves per SIMD, own count removed by one-hot selects or through a register
    index, group decisions per lane or on the scalar unit, label mask constant or per document -- against the C oracle
    (masked cases through the DENSE-layout kernel: sparse_labels=False)."""
    from lda_thesis_amd.sampler import GibbsSampler
    rng = np.random.default_rng(K * 13 + 5)
    D, V = 120, 700
    doc_off, word, freq, labs, z = synth(rng, D, V, K, 1, 70, dense)
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=123, doc_base=50, commit_log=True,
                     sparse_labels=False)
    assert s.commit_log is not None
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(3):
        s.sweep()
        cs.sweep(1, 123, i, doc_base=50, threads=4)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
        np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
    s.check_status()


def test_reciprocal_division_equals_ieee_division():
    """the kernel's p = w/S shortcut (one IEEE reciprocal + two exact-residual corrections) must be
    the correctly rounded quotient: 4e9 random pairs against the hardware division."""
    from lda_thesis_amd import _native
    assert _native.selftest_div(4_000_000_000, seed=7) == 0


def test_two_tier_draw_equals_exact_tier_on_a_large_dense_workload():
    """production margin vs all-exact on 2e5 sites x 4 sweeps (K=512 dense, Zipf words): identical
    assignments and counts; and the production run must actually have used tier 1 (status bit 1 only set
    by exact-tier visits, which are ~1e-9 per site)."""
    import torch
    from lda_thesis_amd.corpus import synthetic_corpus
    from lda_thesis_amd.sampler import GibbsSampler
    off, w, f, z = synthetic_corpus(1000, 200, 20000, 512, seed=9, device="cuda")
    runs = []
    for margin in (0, -1, -2):
        s = GibbsSampler(off, w, f, z, 512, 20000, 0.1, 0.01, labs=None, seed=5)
        s.debug_margin = margin
        for _ in range(4):
            s.sweep()
        runs.append((s.z.clone(), s.n_kw.clone(), s.n_dk.clone(), int(s.status[0].item())))
    for r in runs[1:]:
        assert torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) and torch.equal(runs[0][2], r[2])
    assert runs[1][3] & 2 and not (runs[0][3] & 1)
    # masked variant (K=392-like sparse masks go through the non-dense FAST kernel)
    labs = (torch.rand((1000, 392), device="cuda") < 0.02).cpu().numpy().astype("uint8")
    labs[:, 0] = 1
    import numpy as np
    rng = np.random.default_rng(0)
    zz = np.concatenate([rng.choice(np.nonzero(labs[d])[0], size=200) for d in range(1000)])
    runs = []
    for margin in (0, -1):
        s = GibbsSampler(off, w, f, zz, 392, 20000, 0.1, 0.01, labs=labs, seed=6)
        s.debug_margin = margin
        for _ in range(4):
            s.sweep()
        runs.append((s.z.clone(), s.n_kw.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])


def _run_vs_c(c_oracle, doc_off, word, freq, labs, z, K, V, alpha=0.1, beta=0.01, sweeps=2, **kw):
    from lda_thesis_amd.sampler import GibbsSampler
    s = GibbsSampler(doc_off, word, freq, z, K, V, alpha, beta, labs=labs, seed=7, **kw)
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, alpha, beta)
    for i in range(sweeps):
        s.sweep()
        cs.sweep(1, 7, i, doc_base=kw.get("doc_base", 0), threads=2)
    np.testing.assert_array_equal(s.z_topics(), cs.z)
    np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
    np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
    np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
    s.check_status()
    return s


def _rows16_corpus(K, seed):
    """dense-mask corpus whose words straddle the 16-bit boundary: word 0 and word 3 far above 65535 tokens with all of
    their sites in ONE topic at the start (entries of int32 rows above 2^16 and above 2^24), word 1 with exactly 65535
    tokens in one topic (the largest 16-bit entry), word 2 with 65536 (the first total that must stay int32), the rest rare."""
    rng = np.random.default_rng(seed)
    D, V = 260, 90
    lens = rng.integers(1, 70, size=D)
    lens[:3] = (1, 2, 64)
    doc_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    S = int(doc_off[-1])
    p = 1.0 / np.arange(1, V + 1) ** 1.2
    word = rng.choice(V, size=S, p=p / p.sum()).astype(np.int32)
    word[:8] = (0, 1, 2, 3, 0, 1, 2, 3)
    freq = rng.integers(1, 6, size=S).astype(np.int32)
    z = rng.integers(0, K, size=S).astype(np.int64)
    for w, total, topic in ((0, 20_000_000, 3), (1, 65535, K - 1), (2, 65536, 17), (3, 400_000, 8)):
        idx = np.nonzero(word == w)[0]
        freq[idx] = 1
        freq[idx[0]] = 1 + (total - len(idx)) % (1 << 22)                # (frequencies stay below MAX_FREQ = 2^23)
        rest = total - int(freq[idx].sum())
        j = 1
        while rest > 0:
            add = min(rest, (1 << 22))
            freq[idx[j % len(idx)]] += add
            rest -= add
            j += 1
        assert int(freq[idx].sum()) == total and int(freq[idx].max()) < (1 << 23)
        z[idx] = topic
    return doc_off, word, freq, z, V


@pytest.mark.parametrize("margin", [0, -1, 6])
@pytest.mark.parametrize("K", [512, 1024])
def test_16_bit_rows_equal_the_c_oracle(c_oracle, K, margin):
    """llda_sweep_args.n_kw16: rows of n_kw read from their 16-bit image (llda_pack_rows16) next to int32 rows in the same
    wavefront -- against the C oracle (LabeledLDA.py:106-125), three sweeps, ragged documents, every draw tier."""
    from lda_thesis_amd.sampler import GibbsSampler
    doc_off, word, freq, z, V = _rows16_corpus(K, K + margin)
    labs = np.ones((len(doc_off) - 1, K), dtype=np.uint8)
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=7, commit_log=True, rows16=True, doc_base=11)
    assert s.n_kw16 is not None
    flagged = s.row16.cpu().numpy().astype(bool)
    assert not flagged[0] and flagged[1] and not flagged[2] and not flagged[3] and flagged[4:].all()
    assert int((s.csc_pos < 0).sum()) == int(flagged[word].sum()) > 0
    assert int(s.n_kw.max()) > (1 << 24)
    s.debug_margin = margin
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(3):
        s.sweep()
        cs.sweep(1, 7, i, doc_base=11, threads=4)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
        np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
    s.check_status()
    # the image really is what the kernel read: the flagged rows of the LAST sweep's start, 16 bits per count
    r = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=7, commit_log=True, rows16=False, doc_base=11)
    assert r.n_kw16 is None and int((r.csc_pos < 0).sum()) == 0
    for i in range(3):
        r.sweep()
    import torch
    assert torch.equal(r.z, s.z) and torch.equal(r._counts, s._counts) and torch.equal(r.n_dk, s.n_dk)


def _rows16_corpus_short_docs(K, seed):
    """dense-mask corpus for the FOUR-wave form of the 16-bit-row kernel: every document below 2^16 tokens (n_dk and its
    sweep-start value share an LDS word there), words 0 and 3 above 65535 tokens in total (int32 rows, entries above 2^16 with
    all of word 0 in one topic at the start), word 1 just below, the rest rare; ragged documents from one site to 70."""
    rng = np.random.default_rng(seed)
    D, V = 900, 120
    lens = rng.integers(1, 71, size=D)
    lens[:3] = (1, 2, 70)
    doc_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    S = int(doc_off[-1])
    p = 1.0 / np.arange(1, V + 1) ** 1.1
    word = rng.choice(V, size=S, p=p / p.sum()).astype(np.int32)
    word[:8] = (0, 1, 2, 3, 0, 1, 2, 3)
    freq = rng.integers(1, 6, size=S).astype(np.int32)
    hot = (word == 0) | (word == 3)
    freq[hot] = rng.integers(200, 900, size=int(hot.sum())).astype(np.int32)
    z = rng.integers(0, K, size=S).astype(np.int64)
    z[word == 0] = 3
    i1 = np.nonzero(word == 1)[0]
    freq[i1] = 1
    freq[i1[0]] += 65535 - int(freq[i1].sum())                            # word 1: exactly 65535 tokens, in one document ...
    z[i1] = K - 1
    doc_of = np.searchsorted(doc_off, np.arange(S), side="right") - 1
    tokens = np.bincount(doc_of, weights=freq, minlength=D)
    assert tokens.max() < 65536 + 65535                                   # ... whose other sites are trimmed below
    for d in np.nonzero(tokens >= 65536)[0]:
        sl = slice(doc_off[d], doc_off[d + 1])
        other = np.nonzero(word[sl] != 1)[0] + doc_off[d]
        freq[other] = 1
    tokens = np.bincount(doc_of, weights=freq, minlength=D)
    if tokens.max() >= 65536:                                             # word 1's big site itself: give it a document of its own size
        freq[i1[0]] = 60000
        freq[i1[1 % len(i1)]] += 5535 if len(i1) > 1 else 0
    tokens = np.bincount(doc_of, weights=freq, minlength=D)
    assert tokens.max() < 65536
    assert int(freq[word == 0].sum()) > 65535 and int(freq[word == 3].sum()) > 65535
    return doc_off, word, freq, z, V


@pytest.mark.parametrize("margin", [0, -1, -2, 6])
@pytest.mark.parametrize("K,quad", [(512, True), (512, False), (1024, None), (256, True), (128, True), (100, True), (400, True)])
def test_16_bit_rows_four_waves_equal_the_c_oracle(c_oracle, K, quad, margin):
    """documents below 2^16 tokens (llda_sweep_args.max_doc_tokens) against the C oracle (LabeledLDA.py:106-125), every draw tier:
    K = 512 with FOUR documents per wavefront (quad: the image holds every row, the library flags per sweep the rows that fit -- words 0
    and 3 do not: their sites read the int32 row) and with two (the static flags of llda_pack_rows16), K = 1024 with one -- and, at
    production margins, against the three-wave form (debug_margin -8).  K = 256 and 128: the quad kernel with eight and sixteen
    documents per wavefront (packed own-count removal, 16-byte site records) -- at production margins against the general kernel."""
    import torch
    from lda_thesis_amd.sampler import GibbsSampler
    doc_off, word, freq, z, V = _rows16_corpus_short_docs(K, 3 * K + margin)
    labs = np.ones((len(doc_off) - 1, K), dtype=np.uint8)
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=9, commit_log=True, rows16=True, doc_base=5, quad=quad)
    assert s.n_kw16 is not None and 0 < s.max_doc_tokens < 65536 and s.quad == bool(quad)
    if not s.quad:
        flagged = s.row16.cpu().numpy().astype(bool)
        assert not flagged[0] and not flagged[3] and flagged[1] and flagged[4:].all()
    assert int(s.n_kw.max()) > (1 << 16)
    s.debug_margin = margin
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(3):
        wide_now = (s.n_kw.max(dim=1).values > 65535).cpu().numpy()
        s.sweep()
        if s.quad:                                                       # the flags the sweep just used: exactly the rows that fit
            np.testing.assert_array_equal(s.row16.cpu().numpy().astype(bool), ~wide_now)
            assert wide_now[0] and not wide_now[1]
        cs.sweep(1, 9, i, doc_base=5, threads=4)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
        np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
    s.check_status()
    if margin == 0:
        two_doc = K in (512, 1024)
        r = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=9, commit_log=True, rows16=two_doc, doc_base=5, quad=False)
        assert (r.n_kw16 is not None) == two_doc
        r.debug_margin = -8 if two_doc else 0                           # the three-wave form, production margins (K < 512: int32 rows)
        for i in range(3):
            r.sweep()
        assert torch.equal(r.z, s.z) and torch.equal(r._counts, s._counts) and torch.equal(r.n_dk, s.n_dk)
        if not s.quad:
            assert torch.equal(r.status[1:3], s.status[1:3])             # the same sites left tier 0 / reached the exact tier


@pytest.mark.parametrize("K", [512, 128])
def test_quad_kernel_hands_over_when_wide_rows_become_common(monkeypatch, K):
    """quad=None: counts that concentrate -- here the most frequent words get an entry beyond 65 535 in the middle of the run -- make more
    than QUAD_MAX_WIDE_SITES of the sites read a row that does not fit the 16-bit image; the sampler notices from the library's own flags
    (asynchronously) and goes over to the two-document kernel (K = 128: to the general kernel on int32 rows).  The states equal those
    of a sampler that ran that kernel all along."""
    import torch
    from lda_thesis_amd.sampler import GibbsSampler
    monkeypatch.setattr(GibbsSampler, "QUAD_CHECK_EVERY", 1)
    doc_off, word, freq, z, V = _rows16_corpus_short_docs(K, 77)
    freq = np.minimum(freq, 5).astype(np.int32)                            # (no wide row to begin with)
    a = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=4, commit_log=True)
    b = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=4, commit_log=True, quad=False)
    assert a.quad and not b.quad and a.site_row is None and (b.site_row is not None) == (K == 512)
    hot = np.argsort(np.bincount(word, minlength=V))[::-1][:6]
    for i in range(12):
        if i == 3:
            for s in (a, b):
                s.add_word_topic_counts(hot, np.full(len(hot), 7), np.full(len(hot), 70000))
        a.sweep()
        b.sweep()
        torch.cuda.synchronize()
        assert torch.equal(a.z, b.z) and torch.equal(a._counts, b._counts) and torch.equal(a.n_dk, b.n_dk), i
        # deterministic: the share of sweep 3's image (the first with the wide rows) is queued in sweep 3 and looked at in sweep 4,
        # whatever the host's timing -- every run of the same corpus takes the same kernels
        assert a.quad == (i < 4), i
    assert not a.quad and (a.site_row is not None) == (K == 512)          # handed over (one check interval after the counts changed)
    assert (a.n_kw16 is not None) == (K == 512)
    a.check_status()
    b.check_status()


def test_16_bit_rows_four_waves_flag_a_wrong_token_bound():
    """a max_doc_tokens that is not a bound (a count of n_dk above 65535 under the four-wave form): status bit 2, check_status raises"""
    from lda_thesis_amd.sampler import GibbsSampler
    doc_off, word, freq, z, V = _rows16_corpus_short_docs(512, 1)
    s = GibbsSampler(doc_off, word, freq, z, 512, V, 0.1, 0.01, labs=None, seed=7, commit_log=True, rows16=True)
    s.sweep()
    s.check_status()
    s.n_dk[2, 7] += 70000
    s.sweep()
    with pytest.raises(RuntimeError, match="16 bits"):
        s.check_status()
    # ... and counts that each fit but add up to more than the packed word may ever have to hold
    s = GibbsSampler(doc_off, word, freq, z, 512, V, 0.1, 0.01, labs=None, seed=7, commit_log=True, rows16=True)
    s.n_dk[2, :3] += 30000
    s.sweep()
    with pytest.raises(RuntimeError, match="16 bits"):
        s.check_status()


def test_16_bit_rows_flag_counts_that_do_not_belong_to_the_corpus():
    """a count above 65535 in a row that is read as 16 bits (impossible for counts built from the corpus): llda_pack_rows16
    sets status bit 2 and check_status raises"""
    from lda_thesis_amd.sampler import GibbsSampler
    doc_off, word, freq, z, V = _rows16_corpus(512, 1)
    s = GibbsSampler(doc_off, word, freq, z, 512, V, 0.1, 0.01, labs=None, seed=7, commit_log=True, rows16=True)
    s.sweep()
    s.check_status()
    s.n_kw[5, 100] = 70000
    s.sweep()
    with pytest.raises(RuntimeError, match="16 bits"):
        s.check_status()


def test_16_bit_rows_share_the_allocation_of_the_counts(c_oracle):
    """n_kw16 is carved out of the allocation that holds [n_kw | n_k]: the row offsets the kernel follows (site_row, in 16-byte
    units from n_kw) are a constant of (V, KP), wherever the caching allocator puts anything.  Built here with a 40 GB spacer
    between two other allocations -- the situation in which two separate tensors used to land more than 32 GB apart."""
    import torch
    from lda_thesis_amd.sampler import GibbsSampler
    K = 512
    doc_off, word, freq, z, V = _rows16_corpus(K, 5)
    labs = np.ones((len(doc_off) - 1, K), dtype=np.uint8)
    a = torch.empty((1 << 20,), dtype=torch.uint8, device="cuda")
    spacer = torch.empty((40 << 30,), dtype=torch.uint8, device="cuda")
    b = torch.empty((1 << 20,), dtype=torch.uint8, device="cuda")
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=7, commit_log=True, rows16=True)
    del spacer
    KP = s.layout.KP
    assert s.n_kw16 is not None and s.site_row is not None
    assert s.n_kw16.data_ptr() - s.n_kw.data_ptr() == (V + 1) * KP * 4
    assert s.n_k.data_ptr() - s.n_kw.data_ptr() == V * KP * 4
    assert s._counts.data_ptr() == s.n_kw.data_ptr() and s._counts.numel() == (V + 1) * KP
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(2):
        s.sweep()
        cs.sweep(1, 7, i, threads=4)
    np.testing.assert_array_equal(s.z_topics(), cs.z)
    np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
    np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
    np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
    s.check_status()
    del a, b


def test_16_bit_rows_follow_counts_added_after_construction(c_oracle):
    """add_word_topic_counts (SubLDA's phantom counts, CascadeLDA.py:382-385) changes row totals: a row that no longer fits 16
    bits leaves the image (row16, bit 31 of csc_pos, site_row are recomputed), so nothing is ever truncated"""
    import torch
    from lda_thesis_amd.sampler import GibbsSampler
    K = 512
    doc_off, word, freq, z, V = _rows16_corpus(K, 6)
    labs = np.ones((len(doc_off) - 1, K), dtype=np.uint8)
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=7, commit_log=True, rows16=True)
    assert bool(s.row16[1]) and bool(s.row16[10])
    s.add_word_topic_counts(np.array([10, 10, 1]), np.array([4, 99, 0]), np.array([40000, 30000, 1]))
    assert not bool(s.row16[10]) and not bool(s.row16[1]) and bool(s.row16[11])
    w = torch.as_tensor(word, device="cuda")
    assert int(((s.csc_pos < 0) & ((w == 10) | (w == 1))).sum()) == 0
    assert int((s.csc_pos < 0).sum()) == int(s.row16[w.long()].sum())
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(2):
        s.sweep()
        cs.sweep(1, 7, i, threads=4)
    s.check_status()
    np.testing.assert_array_equal(s.z_topics(), cs.z)
    np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
    np.testing.assert_array_equal(s.n_zk(), cs.n_zk)


def test_edge_empty_shard_and_empty_documents(c_oracle):
    from lda_thesis_amd.sampler import GibbsSampler
    # a shard without documents: sweep is a no-op
    s = GibbsSampler(np.array([0]), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int64), 9, 5, 0.1, 0.01,
                     labs=np.zeros((0, 9), dtype=np.uint8))
    s.sweep()
    assert s.n_k_v().sum() == 0 and s.D == 0
    # ragged: empty documents between one-site and long documents
    rng = np.random.default_rng(1)
    lens = np.array([0, 1, 0, 0, 130, 1, 2, 0, 77, 0])
    off = np.concatenate([[0], np.cumsum(lens)])
    word = np.concatenate([np.sort(rng.choice(200, size=n, replace=False)) for n in lens]).astype(np.int32)
    freq = rng.integers(1, 5, size=off[-1]).astype(np.int32)
    labs = np.ones((10, 24), dtype=np.uint8)
    _run_vs_c(c_oracle, off, word, freq, labs, rng.integers(0, 24, size=off[-1]), 24, 200)


@pytest.mark.parametrize("dpg,permute", [(0, False), (3, True), (1, True)])
@pytest.mark.parametrize("K", [512, 256, 128, 100, 400])
def test_edge_quad_kernel_ragged_empty_documents_and_schedules(c_oracle, dpg, permute, K):
    """the four-documents-per-wavefront kernel (K = 512 dense, commit log; K = 256 / 128: eight / sixteen) on a ragged shard: empty
    documents between one-site and long ones, wavefronts whose documents differ in length by two orders of magnitude, a document count that fills neither the
    last wavefront nor the last workgroup, frequencies above 1, several documents per lane group and a processing order of the
    caller's -- against the C oracle (LabeledLDA.py:106-125), three sweeps, production margins / mixed tiers / exact tier"""
    import torch
    from lda_thesis_amd.sampler import GibbsSampler
    rng = np.random.default_rng(17 + dpg)
    V = 400
    lens = np.concatenate([[0, 1, 0, 0, 330, 1, 2, 0, 77, 0, 0, 0, 0, 5], rng.integers(0, 60, size=37)])
    D = len(lens)
    assert D % 4 and D % 8 and D % 16 and D % 32
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    word = np.concatenate([np.sort(rng.choice(V, size=n, replace=False)) for n in lens]).astype(np.int32)
    freq = rng.integers(1, 6, size=int(off[-1])).astype(np.int32)
    z = rng.integers(0, K, size=int(off[-1]))
    labs = np.ones((D, K), dtype=np.uint8)
    s = GibbsSampler(off, word, freq, z, K, V, 0.1, 0.01, labs=None, seed=13, doc_base=7, commit_log=True, docs_per_group=dpg,
                     sort_docs=not permute)
    assert s.quad and s.commit_log is not None
    if permute:
        s.doc_order = torch.from_numpy(rng.permutation(D).astype(np.int32)).cuda()
    cs = c_oracle.CState(off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(3):
        s.debug_margin = (0, 6, -1)[i]
        s.sweep()
        cs.sweep(1, 13, i, doc_base=7, threads=4)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
        np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
    s.check_status()


@pytest.mark.parametrize("K", [1, 2, 8, 9, 1024])
def test_edge_topic_counts(c_oracle, K):
    rng = np.random.default_rng(K)
    D, V = (12, 300) if K < 1024 else (6, 120)
    lens = rng.integers(1, 40, size=D)
    off = np.concatenate([[0], np.cumsum(lens)])
    word = np.concatenate([np.sort(rng.choice(V, size=n, replace=False)) for n in lens]).astype(np.int32)
    freq = rng.integers(1, 3, size=off[-1]).astype(np.int32)
    labs = np.ones((D, K), dtype=np.uint8)
    _run_vs_c(c_oracle, off, word, freq, labs, rng.integers(0, K, size=off[-1]), K, V)


def test_edge_large_frequencies_last_word_and_root_only_documents(c_oracle):
    rng = np.random.default_rng(3)
    D, K, V = 20, 40, 64
    lens = rng.integers(1, V + 1, size=D)
    lens[0] = V                                             # a document holding every word id, incl. V-1
    off = np.concatenate([[0], np.cumsum(lens)])
    word = np.concatenate([np.sort(rng.choice(V, size=n, replace=False)) for n in lens]).astype(np.int32)
    freq = rng.integers(1, 100000, size=off[-1]).astype(np.int32)      # frequencies far above 1
    labs = (rng.random((D, K)) < 0.1).astype(np.uint8)
    labs[:, 0] = 1
    labs[3] = 0; labs[3, 0] = 1                             # root is the only allowed topic
    z = np.concatenate([rng.choice(np.nonzero(labs[d])[0], size=lens[d]) for d in range(D)])
    s = _run_vs_c(c_oracle, off, word, freq, labs, z, K, V, sweeps=3, doc_base=2 ** 31 - 5)   # doc ids cross 2^31
    assert (s.z_topics()[off[3]:off[4]] == 0).all()


def test_edge_tiny_priors_take_the_exact_kernel(c_oracle):
    """alpha, beta below 1e-6 are outside the tiered kernel's preconditions: the general kernel runs."""
    rng = np.random.default_rng(4)
    D, K, V = 30, 70, 90
    lens = rng.integers(1, 30, size=D)
    off = np.concatenate([[0], np.cumsum(lens)])
    word = np.concatenate([np.sort(rng.choice(V, size=n, replace=False)) for n in lens]).astype(np.int32)
    freq = np.ones(off[-1], dtype=np.int32)
    labs = (rng.random((D, K)) < 0.3).astype(np.uint8)
    labs[:, 0] = 1
    z = np.concatenate([rng.choice(np.nonzero(labs[d])[0], size=lens[d]) for d in range(D)])
    _run_vs_c(c_oracle, off, word, freq, labs, z, K, V, alpha=1e-9, beta=1e-8)
    _run_vs_c(c_oracle, off, word, freq, labs, z, K, V, alpha=0.5, beta=3e10)        # V*beta >= 2^40


# ---- sparse-label kernel (one lane per allowed topic) -------------------------------------------------
@pytest.mark.parametrize("image", [0, 8, 16])
@pytest.mark.parametrize("margin", [0, 6, -1])
@pytest.mark.parametrize("name", ["tiny_k40", "tiny_k392", "tiny_k200", "sublda", "tiny_k1100"])
def test_sparse_kernel_matches_reference_o3(name, margin, image):
    """fixtures whose documents allow few topics run through llda_sweep_sparse_kernel; margin 6 sends many sites
    through the in-kernel exact tier (exact_site_wave), -1 every site; image 8 / 16: the gathers read the saturating narrow
    image of n_kw (llda_pack_image) and escape to n_kw where it shows 255 / 65535."""
    g = load_golden(name)
    s = make_sampler(g, image=image)
    if s.live_off is None:
        pytest.skip("label sets too dense for the sparse kernel")
    assert (s.n_kw_img is not None) == bool(image)
    s.debug_margin = margin
    for i in range(int(g["sweeps"])):
        s.sweep()
        assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics(), name)
    s.check_status()


@pytest.mark.parametrize("margin", [-1, 6])
@pytest.mark.parametrize("K", [9, 40, 100, 129, 190, 257, 392, 640, 777, 900, 968, 1024,
                               1031, 1500, 2100, 3000, 7688])      # the last five: wide layouts (wide exact tier, LDS lock)
def test_sparse_kernel_exact_tier_on_every_layout_shape(c_oracle, K, margin):
    _sparse_exact_tier_case(c_oracle, K, margin, 0)


@pytest.mark.parametrize("image", [8, 16])
@pytest.mark.parametrize("K", [9, 129, 392, 777, 1024, 1031, 3000])
def test_sparse_kernel_exact_tier_with_the_narrow_image(c_oracle, K, image):
    """the same with the gathers going through the 8- / 16-bit image: the exact tier sees the escaped counts too"""
    _sparse_exact_tier_case(c_oracle, K, 6, image)


def _sparse_exact_tier_case(c_oracle, K, margin, image):
    """exact_site_wave (the reference's fp64 pipeline run by a whole wavefront in the dense layout, with G, T, tail and
    the leaf-combine schedule as run-time values): every site (margin -1) or a mixture with decided sites inside one
    wavefront (margin 2^-6) of sparse-label documents, for one / two / four / seven (unbalanced tree) / eight leaves,
    with and without a tail -- against the C oracle (LabeledLDA.py:113-119), three sweeps, ragged documents incl.
    one-site ones."""
    from lda_thesis_amd.sampler import GibbsSampler
    rng = np.random.default_rng(K)
    D, V = 150, 400
    doc_off, word, freq, _, _ = synth(rng, D, V, K, 1, 60, True)
    labs = np.zeros((D, K), dtype=np.uint8)
    labs[:, 0] = 1
    for d in range(D):                                 # root + up to K/4 - 1 (at most 20) other labels
        n = int(rng.integers(0, min(20, K // 4 - 1) + 1))
        labs[d, rng.choice(K - 1, size=n, replace=False) + 1] = 1
    z = np.concatenate([rng.choice(np.nonzero(labs[d])[0], size=doc_off[d + 1] - doc_off[d]) for d in range(D)])
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=77, doc_base=3, image=image)
    assert s.live_off is not None                      # the sparse kernel runs
    assert (s.n_kw_img is not None) == bool(image)
    s.debug_margin = margin
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(3):
        s.sweep()
        cs.sweep(1, 77, i, doc_base=3, threads=4)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
        np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
    s.check_status()
    assert int(s.status[2]) > 0                        # sites really went through the exact tier


def _image_corpus(K, seed, D=700, V=150):
    """sparse label sets (root + up to 7 labels) over a corpus whose counts straddle BOTH saturation values of the narrow image:
    word 0 with 20 million tokens (entries far above 65535), word 1 with a few hundred thousand (entries around 65535: some
    above, some below), word 2 around 255, word 3 exactly 255 tokens in one topic and word 4 exactly 65535, the rest rare; ragged
    documents from one site to 70, so batches of 8 sites end everywhere."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, 71, size=D)
    lens[:3] = (1, 2, 70)
    doc_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    S = int(doc_off[-1])
    word = np.concatenate([np.sort(rng.choice(V, size=n, replace=False, p=None)) for n in lens]).astype(np.int32)
    hot = rng.random(S) < 0.5                                          # half of the documents' first words are the hot ones
    first = np.zeros(S, dtype=bool)
    first[doc_off[:-1]] = True
    word[first & hot] = rng.integers(0, 5, size=int((first & hot).sum()))
    for d in range(D):                                                 # keep word ids unique and ascending inside a document
        sl = slice(doc_off[d], doc_off[d + 1])
        w = np.unique(word[sl])
        while len(w) < lens[d]:
            w = np.unique(np.concatenate([w, rng.integers(5, V, size=lens[d] - len(w))]))
        word[sl] = w
    freq = rng.integers(1, 4, size=S).astype(np.int32)
    labs = np.zeros((D, K), dtype=np.uint8)
    labs[:, 0] = 1
    for d in range(D):
        labs[d, rng.choice(K - 1, size=int(rng.integers(0, 8)), replace=False) + 1] = 1
    doc_of = np.searchsorted(doc_off, np.arange(S), side="right") - 1
    z = np.array([rng.choice(np.nonzero(labs[d])[0]) for d in doc_of], dtype=np.int64)
    for w, total in ((0, 20_000_000), (1, 400_000), (2, 2_000), (3, 255), (4, 65535)):
        idx = np.nonzero(word == w)[0]
        assert len(idx) >= 4, (w, len(idx))
        freq[idx] = 1
        rest = total - len(idx)
        j = 0
        while rest > 0:
            add = min(rest, (1 << 22))
            freq[idx[j % len(idx)]] += add
            rest -= add
            j += 1
        assert int(freq[idx].sum()) == total and int(freq[idx].max()) < (1 << 23)
        if w in (3, 4):
            z[idx] = 0                                                 # the whole word in the root topic: one entry == SAT
    return doc_off, word, freq, z, labs, V


@pytest.mark.parametrize("image", [8, 16])
@pytest.mark.parametrize("K", [64, 392, 512, 1500])
def test_narrow_image_escapes_equal_the_c_oracle(c_oracle, K, image):
    """llda_sweep_args.n_kw_img: counts below, at and above the saturation value of the image next to each other in one wavefront
    -- 255 / 65535 exactly, 20 million -- against the C oracle (LabeledLDA.py:106-125), three sweeps, production margins and the
    exact tier; and bit for bit the sampler that gathers from n_kw itself."""
    import torch
    from lda_thesis_amd.sampler import GibbsSampler
    doc_off, word, freq, z, labs, V = _image_corpus(K, 5 * K + image)
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=21, doc_base=9, image=image)
    r = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=21, doc_base=9, image=0)
    assert s.live_off is not None and s.n_kw_img is not None and r.n_kw_img is None
    sat = 255 if image == 8 else 65535
    assert int((s.n_kw == sat).sum()) >= 1 and int((s.n_kw > sat).sum()) >= 1 and int(s.n_kw.max()) > (1 << 22)
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(3):
        s.debug_margin = r.debug_margin = (0, 6, -1)[i]
        s.sweep()
        r.sweep()
        cs.sweep(1, 21, i, doc_base=9, threads=4)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
        np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
        assert torch.equal(r.z, s.z) and torch.equal(r._counts, s._counts) and torch.equal(r.n_dk, s.n_dk)
    s.check_status()
    # the image the NEXT sweep would read is the saturated copy of the counts
    import lda_thesis_amd._native as nat
    nat.pack_image(s.n_kw, s.n_kw_img)
    img = s.n_kw_img.to(torch.int32) & (0xff if image == 8 else 0xffff)
    assert torch.equal(img, s.n_kw.reshape(-1).clamp(max=sat))


@pytest.mark.parametrize("image", [8, 16])
@pytest.mark.parametrize("K", [512, 1500])
def test_narrow_image_column_order_changes_no_result(c_oracle, K, image):
    """llda_sweep_args.img_col / llda_pack_image_cols: label sets with structure (every document draws its labels from one of a few
    families whose topic ids are scattered over the row) -- the sampler reorders the image's columns so that a family shares cache
    lines; the states with the order forced, refused and chosen by the sampler's own rule are the C oracle's (LabeledLDA.py:106-125),
    every draw tier; and the image is the saturated copy of the counts in that order."""
    import torch
    import lda_thesis_amd._native as nat
    from lda_thesis_amd.sampler import GibbsSampler
    rng = np.random.default_rng(K + image)
    D, V, fam = 700, 300, 14
    ids = rng.permutation(K - 1) + 1
    families = [ids[i * fam:(i + 1) * fam] for i in range((K - 1) // fam)]
    lens = rng.integers(1, 40, size=D)
    doc_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    word = np.concatenate([np.sort(rng.choice(V, size=n, replace=False)) for n in lens]).astype(np.int32)
    freq = rng.integers(1, 4, size=int(doc_off[-1])).astype(np.int32)
    labs = np.zeros((D, K), dtype=np.uint8)
    labs[:, 0] = 1
    for d in range(D):
        f = families[int(rng.integers(len(families)))]
        labs[d, rng.choice(f, size=int(rng.integers(1, 8)), replace=False)] = 1
    z = np.concatenate([rng.choice(np.nonzero(labs[d])[0], size=lens[d]) for d in range(D)])
    runs = {}
    for order in (True, False, None):
        s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=31, doc_base=3, image=image, image_order=order)
        assert s.live_off is not None and s.n_kw_img is not None
        assert (s._img_src is not None) == (order is not False)          # (None: the rule takes it -- the families are scattered)
        if order is not False:
            before, after = s.image_lines_per_site
            assert after < 0.75 * before
            src = s._img_src.cpu().numpy()
            assert sorted(src.tolist()) == list(range(s.layout.KP))
            np.testing.assert_array_equal(s._img_col.cpu().numpy()[src], np.arange(s.layout.KP))
        cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
        for i in range(3):
            s.debug_margin = (0, 6, -1)[i]
            s.sweep()
            cs.sweep(1, 31, i, doc_base=3, threads=4)
            np.testing.assert_array_equal(s.z_topics(), cs.z)
            np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
            np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
        s.check_status()
        runs[order] = (s.z.clone(), s._counts.clone(), s.n_dk.clone())
        if order is True:
            sat = 255 if image == 8 else 65535
            nat.pack_image_cols(s.n_kw, K, s._img_src, s.n_kw_img)
            img = (s.n_kw_img.to(torch.int32) & (0xff if image == 8 else 0xffff)).view(V, s.layout.KP)
            assert torch.equal(img, s.n_kw[:, s._img_src.long()].clamp(max=sat))
    for order in (False, None):
        assert all(torch.equal(a, b) for a, b in zip(runs[True], runs[order]))


def test_sparse_and_dense_kernels_agree_on_a_large_sparse_workload(c_oracle):
    import torch
    from lda_thesis_amd.corpus import synthetic_corpus
    from lda_thesis_amd.sampler import GibbsSampler
    D, N, V = 3000, 120, 5000
    off, w, f, _ = synthetic_corpus(D, N, V, 8, seed=2, device="cuda")
    rng = np.random.default_rng(0)
    for K, nlab in ((392, 7), (512, 15), (100, 20), (2048, 7)):        # (the last: a wide layout)
        labs = np.zeros((D, K), dtype=np.uint8)
        labs[:, 0] = 1
        for d in range(D):
            labs[d, rng.choice(K - 1, size=int(rng.integers(0, nlab + 1)), replace=False) + 1] = 1
        z = np.concatenate([rng.choice(np.nonzero(labs[d])[0], size=N) for d in range(D)])
        runs = []
        for sparse in (True, False):
            s = GibbsSampler(off, w, f, z, K, V, 0.1, 0.01, labs=labs, seed=11, sparse_labels=sparse)
            assert (s.live_off is not None) == sparse
            for _ in range(3):
                s.sweep()
            s.check_status()
            runs.append((s.z.clone(), s.n_kw.clone(), s.n_dk.clone(), s.n_k.clone()))
        for a, b in zip(*runs):
            assert torch.equal(a, b)
    # and against the C oracle for the last configuration
    cs = c_oracle.CState(off.cpu().numpy(), w.cpu().numpy(), f.cpu().numpy(), z, labs,
                         np.zeros((D, K), dtype=np.int64), np.zeros((K, V), dtype=np.int64), np.zeros(K, dtype=np.int64),
                         V, 0.1, 0.01)
    s0 = GibbsSampler(off, w, f, z, K, V, 0.1, 0.01, labs=labs, seed=11)
    cs.n_d_k[:], cs.n_k_v[:], cs.n_zk[:] = s0.n_d_k(), s0.n_k_v(), s0.n_zk()
    for i in range(3):
        cs.sweep(1, 11, i, threads=8)
    np.testing.assert_array_equal(s.z_topics(), cs.z)


@pytest.mark.parametrize("commit", [True, False, "mixed rows", "pipelined"])
def test_exchange_path_on_one_rank(commit, monkeypatch):
    """the multi-GPU path of sweep() -- commit log folded into the DELTA buffer (or atomics on it), RCCL
    all-reduce of the fused delta buffer, llda_apply_delta -- driven on one GPU with a 1-rank nccl group."""
    import os
    import torch.distributed as dist
    g = load_golden("tiny_k130")
    if commit == "mixed rows":              # words with a frequency mass above 12 keep int32 rows, the others int16 pairs
        from lda_thesis_amd.sampler import GibbsSampler
        monkeypatch.setattr(GibbsSampler, "PAIR_LIMIT", 12)
    made = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1)
        made = True
    try:
        s = make_sampler(g, commit_log=bool(commit), exchange_always=True, overlap_ranges=3 if commit == "pipelined" else 1)
        assert (s.rows is not None) == bool(commit)    # int16-pair exchange rows whenever every rank folds a log
        if commit == "pipelined":                      # the rows of a document range travel (async RCCL all-reduce)
            assert len(s._rows_list) == 3              # while the next range is sampled
            s.comm_events = []
        if commit == "mixed rows":
            assert 0 < int((s.row_off < 0).sum()) < s.V
        for i in range(int(g["sweeps"])):
            s.sweep()
            assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics())
            assert int(s._delta.abs().sum()) == 0                  # folded and cleared
        s.check_status()
        if commit == "pipelined":
            assert all(int(r.abs().sum()) == 0 for r in s._rows_list)
            st = s.comm_stats()
            assert st["collectives_per_sweep"] == 3 and st["exposed_ms_per_sweep"] >= 0.0
    finally:
        if made:
            dist.destroy_process_group()


@pytest.mark.parametrize("K,dense", [(1024, True), (512, True), (128, True), (777, False), (257, False), (40, False),
                                     # wide layouts: the fp32 tier of kernel_wide.hpp (margin 112 * 2^-24, bound 92 * 2^-24)
                                     (2048, True), (1088, True), (1031, False), (3000, False), (4296, True)])
def test_adversarial_near_ties_for_the_fp32_tier(c_oracle, K, dense):
    """tests/neartie.py: the last site of every document has its keyed threshold within 2^-24 of a prefix-sum
    boundary (far below fp32 resolution), all other sites are at least 2^-14 away.  The fp32 tier (margin 2^-17,
    proven error bound 105 * 2^-24) must call exactly the tuned sites undecidable -- were its
    rounding error ever above the margin it would decide some of them itself, wrongly half of the time -- and the
    sweep must leave what the exact pipeline (debug_margin = -1) and the C oracle leave
    (/root/reference/LabeledLDA.py:113-119)."""
    import neartie
    from lda_thesis_amd.sampler import GibbsSampler
    st = neartie.make_neartie_state(K, 1500 if K <= 1024 else 400, dense, seed=4242, rng=np.random.default_rng(K), doc_base=7)
    assert st["tuned_gap_max"] < 2.0 ** -23 and st["safe_gap_min"] > 2.0 ** -14.5
    counts = dict(n_d_k=st["n_d_k"], n_k_v=st["n_k_v"], n_zk=st["n_zk"])
    labs = None if dense else st["labs"]
    runs = {}
    margins = (0, -1) + ((-6, -7) if K > 1024 else ())     # wide layouts: the fp32 tier in both of its LDS forms
    for margin in margins:
        s = GibbsSampler(st["doc_off"], st["word"], st["freq"], st["z"], K, st["V"], st["alpha"], st["beta"],
                         labs=labs, counts=counts, seed=4242, doc_base=7, sparse_labels=False)
        s.debug_margin = margin
        s.sweep()
        s.check_status()
        runs[margin] = (s.z_topics(), s.n_d_k(), s.n_k_v(), s.n_zk(), s.status.cpu().numpy())
    cs = c_oracle.CState(st["doc_off"], st["word"], st["freq"], st["z"], st["labs"], st["n_d_k"], st["n_k_v"],
                         st["n_zk"], st["V"], st["alpha"], st["beta"])
    cs.sweep(1, 4242, 0, doc_base=7, threads=4)
    for margin in margins:
        z, ndk, nkv, nzk, _ = runs[margin]
        np.testing.assert_array_equal(z, cs.z)
        np.testing.assert_array_equal(ndk, cs.n_d_k)
        np.testing.assert_array_equal(nkv, cs.n_k_v)
        np.testing.assert_array_equal(nzk, cs.n_zk)
    # production run: the fp32 tier gave up on every tuned site and on no other
    assert int(runs[0][4][1]) == st["n_tuned"]
    for m in (-6, -7):
        if m in runs:
            assert int(runs[m][4][1]) == st["n_tuned"]


def quad_neartie_state(K, D, wide_every=0, seed=4242):
    """tests/neartie.py for the quad kernels (csrc/kernel_quad.hpp): every count fits the 16-bit image (wide_every = n: every n-th
    document's tuned word has counts beyond 65 535 and one beyond 2^24 -- the int32 read without tier 1), documents below 2^16 tokens,
    and the tuned boundary where tier 0's data-dependent margin has the least room: quad lane 0, the last quad lane, the seam
    between chain A and chain B / between two lanes, the last but one position.  Thresholds within 2^-28 of the boundary: the
    margin exceeds the derived error bound by at least 0.125 * 2^-24 of the total, so a tie this close MUST be called undecidable."""
    import neartie
    return neartie.make_neartie_state(K, D, True, seed=seed, rng=np.random.default_rng(K + 7 * wide_every), doc_base=7, xcap=65000, xlog=11.0,
                                      nd_big=600, place="mix", wide_every=wide_every, tuned=2.0 ** -28, rounds=10)


@pytest.mark.parametrize("K,D,wide_every", [(512, 500, 0), (512, 300, 5), (256, 500, 0), (128, 700, 0), (128, 300, 5), (100, 600, 0), (400, 400, 0),
                                             (400, 200, 5)])
def test_adversarial_near_ties_for_the_quad_tier0(c_oracle, K, D, wide_every):
    """The kernel bench.py times (llda_sweep_quad_kernel, csrc/kernel_quad.hpp) decides 99.8 % of its sites in fp32 with a DATA-DEPENDENT
    margin m = 1.05 v (32 L + 39 t + 37 P + 0.25 total) derived by hand against the bound v (31.1 L + 38.2 t + 36.1 P + 0.125 total):
    here every document's last site has its keyed threshold within 2^-28 of a prefix-sum boundary -- in quad lane 0 (P = 0: the margin
    at its smallest), in the last lane, at the chain A / chain B seam, in front of the last slot, a quarter of the documents each --
    and all other sites are at least 2^-14.5 away.  Tier 0 must hand over EVERY tuned site and no other: with the production margin,
    with the margin scaled down to the derived bound itself (debug_margin -10) and with the constant margin (-9); the sweep must
    leave what the exact pipeline (-1) and the C oracle leave (/root/reference/LabeledLDA.py:113-119).  K = 100, 400: positions
    without a topic (PAD); wide_every: rows the 16-bit image cannot hold (int32 read, + 2 v, no tier 1)."""
    from lda_thesis_amd.sampler import GibbsSampler
    st = quad_neartie_state(K, D, wide_every)
    assert st["tuned_gap_max"] < 2.0 ** -27.5 and st["safe_gap_min"] > 2.0 ** -14.5
    assert int(st["n_d_k"].sum(axis=1).max()) < 65536
    last_word = st["word"][st["doc_off"][1:] - 1]
    assert int(st["n_k_v"][:, np.setdiff1d(np.arange(st["V"]), last_word[st["wide_docs"]])].max()) <= 65535
    placed = np.bincount(st["place_of"][st["place_of"] >= 0], minlength=4)
    assert (placed >= D // 16).all(), placed                 # every kind of boundary is there (each is wanted by a quarter of the documents)
    counts = dict(n_d_k=st["n_d_k"], n_k_v=st["n_k_v"], n_zk=st["n_zk"])
    runs = {}
    for margin in (0, -10, -9, -1):
        s = GibbsSampler(st["doc_off"], st["word"], st["freq"], st["z"], K, st["V"], st["alpha"], st["beta"],
                         counts=counts, seed=4242, doc_base=7, commit_log=True, rows16=True, quad=True)
        assert s.quad and s.row16 is not None
        s.debug_margin = margin
        s.sweep()
        s.check_status()
        if wide_every:
            assert int((s.row16 == 0).sum()) == int(st["wide_docs"].sum())       # exactly the planted rows are read as int32
        else:
            assert int((s.row16 == 0).sum()) == 0
        runs[margin] = (s.z_topics(), s.n_d_k(), s.n_k_v(), s.n_zk(), s.status.cpu().numpy())
    cs = c_oracle.CState(st["doc_off"], st["word"], st["freq"], st["z"], st["labs"], st["n_d_k"], st["n_k_v"],
                         st["n_zk"], st["V"], st["alpha"], st["beta"])
    cs.sweep(1, 4242, 0, doc_base=7, threads=4)
    for margin, (z, ndk, nkv, nzk, _) in runs.items():
        np.testing.assert_array_equal(z, cs.z, err_msg="debug_margin %d" % margin)
        np.testing.assert_array_equal(ndk, cs.n_d_k)
        np.testing.assert_array_equal(nkv, cs.n_k_v)
        np.testing.assert_array_equal(nzk, cs.n_zk)
    # tier 0 gave up on every tuned site and on no other
    for margin in (0, -10, -9):
        assert int(runs[margin][4][1]) == st["n_tuned"], (margin, int(runs[margin][4][1]), st["n_tuned"])
    # ... and the fp64 decision behind it (margin 2^-40: in the quad layout, or -- int32 rows -- in the standard layout out of line)
    # settled most of them (the tuner stops below 2^-28 and often lands far below: a few per cent are within 2^-40, the exact tier's)
    assert int(runs[0][4][2]) <= max(2, st["n_tuned"] // 10)
    assert int(runs[-1][4][2]) == int(st["doc_off"][-1])          # debug_margin -1: every site through the exact pipeline


@pytest.mark.parametrize("K,labels,image", [(40, 7, 0), (392, 7, 0), (512, 7, 8), (512, 20, 16), (777, 40, 0), (1031, 7, 0), (2048, 7, 8)])
def test_adversarial_near_ties_for_the_sparse_fp32_tier(c_oracle, K, labels, image):
    """the same for the sparse-label kernel's tier 0 (kernel_sparse.hpp: fp32 scores of the allowed topics, margin 2^-17, bound
    41 * 2^-24): the last site of every document within 2^-24 of a prefix-sum boundary, all others at least 2^-14.5 away.  Tier 0
    must hand exactly the tuned sites to the fp64 decision, and the sweep must leave what the exact pipeline and the C oracle leave
    (/root/reference/LabeledLDA.py:113-119) -- with and without the narrow image (8, 16 and 64 lanes per document; wide layouts)."""
    import neartie
    from lda_thesis_amd.sampler import GibbsSampler
    st = neartie.make_neartie_state(K, 1200 if K <= 1024 else 400, False, seed=4242, rng=np.random.default_rng(K + labels), doc_base=7,
                                    label_count=labels)
    assert st["tuned_gap_max"] < 2.0 ** -23 and st["safe_gap_min"] > 2.0 ** -14.5
    counts = dict(n_d_k=st["n_d_k"], n_k_v=st["n_k_v"], n_zk=st["n_zk"])
    runs = {}
    for margin in (0, -1):
        s = GibbsSampler(st["doc_off"], st["word"], st["freq"], st["z"], K, st["V"], st["alpha"], st["beta"],
                         labs=st["labs"], counts=counts, seed=4242, doc_base=7, image=image)
        assert s.live_off is not None and (s.n_kw_img is not None) == bool(image)
        s.debug_margin = margin
        s.sweep()
        s.check_status()
        runs[margin] = (s.z_topics(), s.n_d_k(), s.n_k_v(), s.n_zk(), s.status.cpu().numpy())
    cs = c_oracle.CState(st["doc_off"], st["word"], st["freq"], st["z"], st["labs"], st["n_d_k"], st["n_k_v"],
                         st["n_zk"], st["V"], st["alpha"], st["beta"])
    cs.sweep(1, 4242, 0, doc_base=7, threads=4)
    for margin in (0, -1):
        z, ndk, nkv, nzk, _ = runs[margin]
        np.testing.assert_array_equal(z, cs.z)
        np.testing.assert_array_equal(ndk, cs.n_d_k)
        np.testing.assert_array_equal(nkv, cs.n_k_v)
        np.testing.assert_array_equal(nzk, cs.n_zk)
    # production run: the fp32 tier gave up on every tuned site and on no other; the fp64 decision behind it settled (nearly) all of
    # them -- a tuned threshold may sit within 2^-40 of its boundary, which is the exact pipeline's to decide
    assert int(runs[0][4][1]) == st["n_tuned"] and int(runs[0][4][2]) <= st["n_tuned"] // 100
    assert int(runs[-1][4][2]) == int(st["doc_off"][-1])          # debug_margin -1: every site through the exact pipeline


@pytest.mark.parametrize("K,image,heavy", [(392, 0, False), (512, 8, False), (2048, 8, False), (392, 0, True), (512, 8, True), (100, 0, True),
                                           (1100, 16, True)])
def test_sparse_documents_are_split_by_the_lanes_they_need(c_oracle, K, image, heavy):
    """GibbsSampler._lane_parts: a corpus in which most documents allow a handful of topics and a few allow up to 64 runs as one launch of
    the sparse-label kernel per class of lanes (8 / 16 / 32 / 64 per document) instead of giving every document 64 lanes; `heavy`: a few
    documents allow more than 64 topics / more than a quarter of K -- they alone take the dense kernel with their label masks (before,
    one such document sent the whole corpus there).  The state is the C oracle's (LabeledLDA.py:106-125) and bit for bit that of the
    dense kernel on everything."""
    import torch
    from lda_thesis_amd.sampler import GibbsSampler
    rng = np.random.default_rng(K + heavy)
    D, V = 600, 500
    doc_off, word, freq, _, _ = synth(rng, D, V, K, 0, 50, True)
    labs = np.zeros((D, K), dtype=np.uint8)
    labs[:, 0] = 1
    top = min(63, K // 4 - 1)
    sizes = [3, 7, min(12, top), min(15, top), min(25, top), min(31, top), min(40, top), top]
    n_lab = rng.choice(sizes, size=D, p=[0.4, 0.3, 0.08, 0.07, 0.05, 0.04, 0.03, 0.03])
    if heavy:
        n_lab[rng.choice(D, size=25, replace=False)] = rng.choice([K // 4 + 1, min(K - 1, 70), K - 1], size=25)
    for d in range(D):
        labs[d, rng.choice(K - 1, size=n_lab[d], replace=False) + 1] = 1
    z = np.concatenate([rng.choice(np.nonzero(labs[d])[0], size=doc_off[d + 1] - doc_off[d]) for d in range(D)])
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=5, doc_base=2, image=image)
    one = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=labs, seed=5, doc_base=2, sparse_labels=False)
    assert s.live_off is not None and one.live_off is None
    parts = s._lane_parts(0, s.D, s.doc_order)
    classes = [p[2] for p in parts]
    assert classes == sorted(set(classes) - {0}) + ([0] if heavy else []) and len(classes) >= 3 and sum(p[1] for p in parts) == D
    assert sorted(torch.cat([p[0] for p in parts]).cpu().tolist()) == list(range(D))
    if heavy:
        assert parts[-1][1] == 25 and (s._heavy.cpu().numpy() == ((n_lab + 1 > 64) | ((n_lab + 1) * 4 > K))).all()
    cs = c_oracle.CState(doc_off, word, freq, z, labs, s.n_d_k(), s.n_k_v(), s.n_zk(), V, 0.1, 0.01)
    for i in range(3):
        s.debug_margin = one.debug_margin = (0, 6, 0)[i]
        s.sweep()
        one.sweep()
        cs.sweep(1, 5, i, doc_base=2, threads=4)
        np.testing.assert_array_equal(s.z_topics(), cs.z)
        np.testing.assert_array_equal(s.n_d_k(), cs.n_d_k)
        np.testing.assert_array_equal(s.n_k_v(), cs.n_k_v)
        np.testing.assert_array_equal(s.n_zk(), cs.n_zk)
        assert torch.equal(one.z, s.z) and torch.equal(one._counts, s._counts) and torch.equal(one.n_dk, s.n_dk)
    s.check_status()


@pytest.mark.parametrize("bits", [8, 16])
def test_pack_image_cols_clamps_a_bad_column_table(bits):
    """include/llda_gibbs.h: col_src is the caller's table and a PRECONDITION (a permutation of the positions); an entry outside
    0 .. KP-1 must not read someone else's LDS -- the kernel clamps it to KP-1.  A valid table gives the permuted
    saturating image, a table with wild entries gives the image of the clamped table, and nothing faults."""
    import torch
    from lda_thesis_amd import _native as nat
    K, V = 512, 300
    KP = nat.layout_init(K)["KP"]
    g = torch.Generator(device="cuda").manual_seed(3)
    n_kw = torch.randint(0, 70000, (V, KP), dtype=torch.int32, device="cuda", generator=g)
    sat = 255 if bits == 8 else 65535
    img = torch.zeros((V * KP,), dtype=torch.uint8 if bits == 8 else torch.int16, device="cuda")
    perm = torch.randperm(KP, device="cuda", generator=g).to(torch.int32)
    nat.pack_image_cols(n_kw, K, perm, img)
    want = n_kw[:, perm.to(torch.int64)].clamp(max=sat)
    got = img.view(V, KP).to(torch.int32) & sat
    assert torch.equal(got, want)
    bad = perm.clone()
    bad[::7] = torch.tensor([KP, 10 ** 6, -1, -(2 ** 31), 2 ** 31 - 1], dtype=torch.int32, device="cuda").repeat(KP)[:bad[::7].numel()]
    nat.pack_image_cols(n_kw, K, bad, img)
    torch.cuda.synchronize()
    clamped = bad.to(torch.int64)
    clamped = torch.where((clamped < 0) | (clamped > KP - 1), torch.full_like(clamped, KP - 1), clamped)
    got = img.view(V, KP).to(torch.int32) & sat
    assert torch.equal(got, n_kw[:, clamped].clamp(max=sat))
