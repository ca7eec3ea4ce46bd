"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle cannot run 37.5 M sites
in test time): count conservation, counts == a recount from the assignments, independence of the work
schedule and of the commit path, and independence of how the documents are sharded (two half shards with
their deltas summed by hand == one shard) -- which is what makes 1/2/4/8-GPU runs bit-identical.

configs[2]: 100k docs x 200 sites, K=128, V=50k, dense mask; configs[3]: one GPU's 125k docs x 300 sites of the
1M-document corpus, K=512, V=100k, dense mask -- and the WHOLE 1M-document corpus on one GPU (what bench.py times
at N = 1).

On top of the properties, the oracle itself is applied at full size BY SAMPLING: under per-document snapshot
semantics a document depends only on the sweep-start n_kw / n_k and on itself, so a few thousand documents picked
at random are swept by oracle/llda_oracle.c (llda_oracle_sweep_docs, reference LabeledLDA.py:106-125 restated)
against the sweep-start counts and compared bit for bit with what the HIP sweep of the whole corpus left for
them; and the production draw tiers are compared with the fp32 tier switched off (debug_margin = -2) on the full
sweep."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = {"synth1": (100_000, 200, 50_000, 128), "synth_k256": (100_000, 200, 50_000, 256), "synth2_shard": (125_000, 300, 100_000, 512),
           "wide_k2048": (20_000, 100, 20_000, 2048)}      # (a wide layout: bench.py's extra.wide_k2048)


def corpus(name):
    from lda_thesis_amd.corpus import synthetic_corpus, synthetic_corpus_blocks
    if name == "synth2_1M":                                 # bench.py's default workload, generated the same way
        return synthetic_corpus_blocks(0, 1_000_000, 300, 100_000, 512, 1234, "cuda") + (512, 100_000)
    D, N, V, K = CONFIGS[name]
    return synthetic_corpus(D, N, V, K, 1234, "cuda") + (K, V)


def sample_documents(s, n_docs, rng):
    """random documents of the sampler: (ids, local CSR offsets, site indices on the device)."""
    sel = np.sort(rng.choice(s.D, size=n_docs, replace=False))
    sel_t = torch.from_numpy(sel).to(s.device)
    lens = s.doc_off[sel_t + 1] - s.doc_off[sel_t]
    loc_off = torch.zeros(n_docs + 1, dtype=torch.int64, device=s.device)
    torch.cumsum(lens, 0, out=loc_off[1:])
    idx = torch.repeat_interleave(s.doc_off[sel_t] - loc_off[:-1], lens) + torch.arange(int(loc_off[-1]), device=s.device)
    return sel, sel_t, loc_off.cpu().numpy(), idx


def sweep_and_check_sample(s, c_oracle, n_docs, rng):
    """one HIP sweep of everything; the sampled documents must come out exactly as the C oracle leaves them."""
    sel, sel_t, loc_off, idx = sample_documents(s, n_docs, rng)
    n_k_v, n_zk = s.n_k_v(), s.n_zk()                       # sweep-start counts, reference layout
    word, freq = s.word[idx].cpu().numpy(), s.freq[idx].cpu().numpy()
    z0 = s._pos_topic[s.z[idx].to(torch.int64)].cpu().numpy()
    ndk0 = s.n_dk[sel_t][:, s._topic_pos].cpu().numpy().astype(np.int64)
    sweep = s.sweeps_done
    s.sweep()
    z_want, ndk_want = c_oracle.sweep_docs(sel + s.doc_base, loc_off, word, freq, z0, None, ndk0, n_k_v, n_zk, s.V,
                                           s.alpha, s.beta, s.seed, sweep, stream=s.stream_id,
                                           threads=min(32, os.cpu_count() or 1))
    z_got = s._pos_topic[s.z[idx].to(torch.int64)].cpu().numpy()
    ndk_got = s.n_dk[sel_t][:, s._topic_pos].cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(z_got, z_want)
    np.testing.assert_array_equal(ndk_got, ndk_want)
    assert int((z_got != z0).sum()) > z0.size // 2          # the sampled sites really moved


def make(doc_off, word, freq, z, K, V, **kw):
    from lda_thesis_amd.sampler import GibbsSampler
    return GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=None, counts=None, seed=42, **kw)


def check_conservation(s):
    total = int(s.freq.sum(dtype=torch.int64))
    assert int(s.n_kw.sum(dtype=torch.int64)) == total and int(s.n_k.sum(dtype=torch.int64)) == total
    assert torch.equal(s.n_kw.sum(dim=0, dtype=torch.int64), s.n_k.to(torch.int64))          # n_k = column sums
    assert torch.equal(s.n_dk.sum(dim=0, dtype=torch.int64), s.n_k.to(torch.int64))          # ... of n_dk too
    per_doc = torch.zeros((s.D,), dtype=torch.int64, device=s.device)
    rows = torch.repeat_interleave(torch.arange(s.D, device=s.device), s.doc_off[1:] - s.doc_off[:-1])
    per_doc.index_add_(0, rows, s.freq.to(torch.int64))
    assert torch.equal(s.n_dk.sum(dim=1, dtype=torch.int64), per_doc)
    assert int(s.n_kw.min()) >= 0 and int(s.n_dk.min()) >= 0


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_properties(name):
    doc_off, word, freq, z, K, V = corpus(name)
    s = make(doc_off, word, freq, z, K, V)
    assert s.commit_log is not None                        # production path at this size: word-major commit log
    start = s._counts.clone()
    z0 = s.z.clone()
    for _ in range(2):
        s.sweep()
    s.check_status()
    check_conservation(s)
    assert int((s.z != z0).sum()) > s.S // 2               # the chain really moved

    # counts maintained incrementally == counts rebuilt from the final assignments (llda_count_init)
    r = make(doc_off, word, freq, s.z_topics(), K, V)
    assert torch.equal(r.n_kw, s.n_kw) and torch.equal(r.n_dk, s.n_dk) and torch.equal(r.n_k, s.n_k)
    del r

    # a different schedule (documents per workgroup, document order) and the atomics commit path: same state
    t = make(doc_off, word, freq, z, K, V, docs_per_group=3, sort_docs=False, commit_log=False)
    t.doc_order = torch.randperm(t.D, device=t.device, generator=torch.Generator(device=t.device).manual_seed(1)).to(torch.int32)
    for _ in range(2):
        t.sweep()
    assert torch.equal(t.z, s.z) and torch.equal(t.n_kw, s.n_kw) and torch.equal(t.n_dk, s.n_dk) and torch.equal(t.n_k, s.n_k)
    del t

    # two half shards (as two GPUs would hold them), deltas summed by hand == the single shard
    from lda_thesis_amd.sampler import shard_documents
    b = shard_documents(doc_off.cpu().numpy(), 2)
    off = doc_off.cpu()
    halves = []
    for r_ in range(2):
        lo, hi = b[r_], b[r_ + 1]
        s0, s1 = int(off[lo]), int(off[hi])
        # exchange path without a process group: every half folds its own rows (int16 pairs for all but the hot words)
        h = make(doc_off[lo:hi + 1] - doc_off[lo], word[s0:s1], freq[s0:s1], z[s0:s1], K, V, doc_base=lo,
                 exchange_always=True, commit_log=True)
        assert h.rows is not None and int((h.row_off < 0).sum()) > V // 2
        halves.append(h)
    counts = start.clone()
    for _ in range(2):
        deltas = []
        for h in halves:
            h._counts.copy_(counts)                        # what the all-reduce + fold would leave on every rank
            h.sweep()
            deltas.append(h._counts - counts)
        counts = counts + deltas[0] + deltas[1]
    assert torch.equal(counts, s._counts)
    assert torch.equal(torch.cat([h.z for h in halves]), s.z)
    assert torch.equal(torch.cat([h.n_dk for h in halves]), s.n_dk)


@pytest.mark.parametrize("name,n_docs", [("synth1", 3000), ("synth_k256", 2500), ("synth2_shard", 2000), ("synth2_1M", 2000), ("wide_k2048", 600)])
def test_full_size_sampled_documents_vs_c_oracle(c_oracle, name, n_docs):
    """the oracle at full size, by sampling (see the module docstring): two sweeps, a fresh sample each."""
    doc_off, word, freq, z, K, V = corpus(name)
    s = make(doc_off, word, freq, z, K, V)
    del z
    assert s.commit_log is not None and s.debug_margin == 0     # production path: tiers + word-major commit log
    if K in (128, 256, 512):
        assert s.n_kw16 is not None and s.quad                   # ... on the kernel bench.py times (K = 512: 4 documents per wavefront,
                                                                 # 256: 8, 128: 16)
    rng = np.random.default_rng(2024)
    for _ in range(2):
        sweep_and_check_sample(s, c_oracle, n_docs, rng)
    s.check_status()
    check_conservation(s)


@pytest.mark.parametrize("name", ["synth1", "synth_k256", "synth2_shard", "synth2_1M"])
def test_full_size_production_margins_vs_no_fp32_tier(name):
    """every site of the full-size sweep: the production tiers (fp32 decision first) pick the topic the fp64 tiers
    pick with the fp32 tier switched off (debug_margin = -2), two sweeps.  synth2_1M is the workload bench.py times, on the
    instantiation it times (K = 512: four documents per wavefront on the 16-bit image, 3 * 10^8 sites)."""
    doc_off, word, freq, z, K, V = corpus(name)
    a = make(doc_off, word, freq, z, K, V)
    b = make(doc_off, word, freq, z, K, V)
    del z
    if K in (128, 256, 512):
        assert a.n_kw16 is not None and a.quad and b.quad
    b.debug_margin = -2
    for _ in range(2):
        a.sweep()
        b.sweep()
        assert torch.equal(a.z, b.z)
    assert torch.equal(a._counts, b._counts) and torch.equal(a.n_dk, b.n_dk)
    st = a.status.cpu().numpy()
    assert 0 < int(st[1]) < a.S // 20                      # the fp32 tier was unsure about some sites, but few
    assert int(b.status[1]) == 2 * b.S                     # ... and b sent every site to the fp64 tiers


def sparse_corpus(docs):
    """the sparse variant of configs[3] (root + 7 random labels per document, as bench.py's synth2_sparse builds it)"""
    from lda_thesis_amd.corpus import synthetic_corpus_blocks
    N, V, K = 300, 100_000, 512
    doc_off, word, freq, _ = synthetic_corpus_blocks(0, docs, N, V, K, 1234, "cuda")
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99)
    lab = torch.sort(torch.randint(1, K - 8, (docs, 7), device="cuda", generator=gen), dim=1).values + torch.arange(7, device="cuda")
    lab = torch.cat([torch.zeros((docs, 1), dtype=lab.dtype, device="cuda"), lab], dim=1)
    pick = torch.randint(0, 8, (docs * N,), device="cuda", generator=gen)
    z = lab.repeat_interleave(N, dim=0)[torch.arange(docs * N, device="cuda"), pick]
    return doc_off, word, freq, z, lab, K, V


@pytest.mark.parametrize("docs,n_docs,bits", [(125_000, 2000, 8), (1_000_000, 1500, 8)])
def test_full_size_sparse_labels_sampled_documents_vs_c_oracle(c_oracle, docs, n_docs, bits):
    """the sparse-label kernel at full size with the image GibbsSampler picks by itself (8 bits: 7 % of the gathers escape to the
    int32 counts at 125 000 documents, 35 % at 1 000 000), fp32 tier 0 first: sampled documents against the C oracle
    (LabeledLDA.py:106-125 restated), two sweeps, a fresh sample each."""
    from lda_thesis_amd.sampler import GibbsSampler
    doc_off, word, freq, z, lab, K, V = sparse_corpus(docs)
    lab_off = np.arange(0, 8 * docs + 1, 8, dtype=np.int64)
    s = GibbsSampler(doc_off, word, freq, z, K, V, 0.1, 0.01, labs=(lab_off, lab.reshape(-1).cpu().numpy()), seed=42)
    del z
    assert s.live_off is not None and s.commit_log is not None and s.n_kw_img is not None
    assert 8 * s.n_kw_img.element_size() == bits
    r8, r16 = s._image_escape_rates()
    assert 0.02 < r8 < s.IMAGE_MAX_ESCAPES and r16 < r8
    rng = np.random.default_rng(7)
    for _ in range(2):
        sel, sel_t, loc_off, idx = sample_documents(s, n_docs, rng)
        n_k_v, n_zk = s.n_k_v(), s.n_zk()
        word_h, freq_h = s.word[idx].cpu().numpy(), s.freq[idx].cpu().numpy()
        z0 = s._pos_topic[s.z[idx].to(torch.int64)].cpu().numpy()
        ndk0 = s.n_dk[sel_t][:, s._topic_pos].cpu().numpy().astype(np.int64)
        labs_sel = np.zeros((n_docs, K), dtype=np.uint8)
        labs_sel[np.repeat(np.arange(n_docs), 8), lab[sel_t].cpu().numpy().reshape(-1)] = 1
        sweep = s.sweeps_done
        s.sweep()
        z_want, ndk_want = c_oracle.sweep_docs(sel, loc_off, word_h, freq_h, z0, labs_sel, ndk0, n_k_v, n_zk, V, 0.1, 0.01, 42, sweep,
                                               threads=min(32, os.cpu_count() or 1))
        z_got = s._pos_topic[s.z[idx].to(torch.int64)].cpu().numpy()
        np.testing.assert_array_equal(z_got, z_want)
        np.testing.assert_array_equal(s.n_dk[sel_t][:, s._topic_pos].cpu().numpy().astype(np.int64), ndk_want)
        assert int((z_got != z0).sum()) > z0.size // 2
    s.check_status()
    check_conservation(s)
    st = s.status.cpu().numpy()
    assert 0 < int(st[1]) < s.S // 100                     # the fp32 tier handed a few sites to the fp64 decision


@pytest.mark.parametrize("docs,N,V,K", [(31_250, 300, 100_000, 512), (50_000, 200, 50_000, 256), (50_000, 200, 50_000, 128)])
def test_quad_kernel_equals_the_two_document_kernel_in_every_tier_mode(docs, N, V, K):
    """K = 512: four documents per wavefront (csrc/kernel_quad.hpp) against the two-document 16-bit-row kernel on a slice of configs[3]
    large enough for every CU to run several workgroups at once -- the full integer state after each of two sweeps, with production
    margins, without the fp32 tier, with margins 2^-6 and with every site through the exact tier (9.4 million sites each).  K = 256 and
    128 (configs[2]): eight and sixteen documents per wavefront against the general kernel on int32 rows.
    (A stale register index in the quad kernel's own-count removal showed only beyond the first workgroup of a CU and only once enough
    registers were live for the stray write to land in one: tools/quad_debug.py.)"""
    from lda_thesis_amd.corpus import synthetic_corpus_blocks
    doc_off, word, freq, z = synthetic_corpus_blocks(0, docs, N, V, K, 1234, "cuda", block=docs // 2)
    ref = None
    for quad, margin in ((False, 0), (True, 0), (True, -2), (True, 6), (True, -1)):
        s = make(doc_off, word, freq, z, K, V, quad=quad)
        assert s.quad == quad and (s.n_kw16 is not None) == (quad or K == 512)
        s.debug_margin = margin
        states = []
        for _ in range(2):
            s.sweep()
            states.append((s.z.clone(), s.n_dk.clone(), s._counts.clone()))
        s.check_status()
        st = s.status.cpu().numpy()
        if margin == -1:
            assert int(st[2]) == 2 * s.S                                  # every site reached the exact tier
        if margin == 0:
            assert 0 < int(st[1]) < s.S // 20 and int(st[2]) < 10
        if ref is None:
            ref = states
        for (a, b) in zip(ref, states):
            assert all(torch.equal(x, y) for x, y in zip(a, b)), (quad, margin)
        del s


def test_exchange_rows_of_configs3_are_what_design_section_7_prices():
    """DESIGN.md section 7 prices the per-sweep all-reduce of BASELINE configs[3] at 103.5 MB: one int32 SUM over the exchange rows of
    the whole 1 M-document corpus -- int16 pairs for every word whose frequency mass over ALL ranks is at most 32 767 (98 920 of the
    100 000 words), int32 for the hot ones, plus the n_k row.  The mass is a property of the corpus, so the size is the same on
    every rank of any N; here on one rank with the exchange forced on (/root/reference/LabeledLDA.py:109-111,123-125 are the updates
    that travel)."""
    doc_off, word, freq, z, K, V = corpus("synth2_1M")
    s = make(doc_off, word, freq, z, K, V, exchange_always=True, commit_log=True)
    del z
    assert s.rows is not None and s.quad
    pairs = int((s.row_off[:-1] < 0).sum())
    assert pairs == 98_920
    assert s.rows.numel() * 4 == (pairs * 256 + (V - pairs) * 512 + 512) * 4 == 103_507_968
    # ... against the 204.8 MB of the plain int32 delta buffer
    assert (V + 1) * 512 * 4 == 204_802_048
    s.sweep()                                              # and the path runs: log -> rows -> counts, rows cleared
    s.check_status()
    assert int(s.rows.abs().sum()) == 0
    check_conservation(s)
