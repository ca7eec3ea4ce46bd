"""Host-side logic that needs no GPU: layout algebra, sharding, scheduling, corpus/label handling."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st


@settings(max_examples=120, deadline=None)
@given(st.one_of(st.integers(1, 1100), st.integers(1101, 7688)))
def test_layout_is_a_permutation_with_aligned_rows(K):
    from lda_thesis_amd.layout import GroupLayout
    L = GroupLayout(K)
    if L.wide:          # more than 8 pairwise leaves: 64-lane tiers of one wavefront
        assert L.m > 8 and L.G == 64 * L.NT and L.NT <= 8 and L.T in (8, 12, 16) and L.KP == L.G * L.T <= 8192
        assert len(L.comb) == L.m - 1 and L.comb[-1][0] == 0
    else:
        assert L.G in (8, 16, 32, 64) and L.T in (1, 2, 4, 8, 12, 16) and L.KP == L.G * L.T
    assert sorted(L.topic_pos.tolist()) == sorted(set(L.topic_pos.tolist())) and L.topic_pos.max() < L.KP
    assert (L.pos_topic[L.topic_pos] == np.arange(K)).all() and (L.pos_topic >= 0).sum() == K
    x = np.arange(1, K + 1)
    np.testing.assert_array_equal(L.from_device(L.to_device(x)), x)
    # chain structure: topic k sits in lane 8*leaf + (rel & 7), slot rel >> 3 ...
    for p, (start, n) in enumerate(L.leaves):
        for rel in (0, n - 1):
            k = start + rel
            assert L.topic_lane[k] == 8 * p + (rel & 7) and L.topic_slot[k] == rel >> 3
            assert L.pos_lane[L.topic_pos[k]] == L.topic_lane[k] and L.pos_slot[L.topic_pos[k]] == L.topic_slot[k]
    # ... and in memory the 16-byte chunk s // 4 of ALL lanes is contiguous (rows of 1 or 2 slots: lane-major)
    g, sl = np.divmod(np.arange(L.KP), L.T)
    want = ((sl // 4) * L.G + g) * 4 + sl % 4 if L.T % 4 == 0 else g * L.T + sl
    np.testing.assert_array_equal(L.lm_pos, want)
    assert sorted(L.lm_pos.tolist()) == list(range(L.KP))
    np.testing.assert_array_equal(L.draw_rank[L.lm_pos], np.arange(L.KP))


def test_lane_masks_roundtrip():
    from lda_thesis_amd.layout import GroupLayout
    rng = np.random.default_rng(0)
    for K in (5, 20, 130, 392, 1000, 1031, 2100):
        L = GroupLayout(K)
        labs = (rng.random((7, K)) < 0.3).astype(np.uint8)
        m = L.lane_masks(labs).astype(np.int64)
        np.testing.assert_array_equal(L.labs_from_masks(m), labs)
        bits = ((m[:, :, None] >> np.arange(16)) & 1)
        assert bits.sum() == labs.sum()                     # nothing set in the padding


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(0, 50), min_size=1, max_size=60), st.integers(1, 8))
def test_shard_documents_covers_everything_in_order(lens, world):
    from lda_thesis_amd.sampler import shard_documents
    off = np.concatenate([[0], np.cumsum(lens)])
    b = shard_documents(off, world)
    assert b[0] == 0 and b[-1] == len(lens) and len(b) == world + 1
    assert all(b[i] <= b[i + 1] for i in range(world))


def test_shard_documents_balances_sites():
    from lda_thesis_amd.sampler import shard_documents
    off = np.arange(0, 8001, 8)
    b = shard_documents(off, 8)
    assert [b[i + 1] - b[i] for i in range(8)] == [125] * 8


def test_lpt_assign():
    from lda_thesis_amd.CascadeLDA import lpt_assign
    costs = [200, 80, 50, 49, 30, 30, 10, 5]
    owner = lpt_assign(costs, 3)
    loads = [sum(c for c, o in zip(costs, owner) if o == w) for w in range(3)]
    assert max(loads) == 200 and sorted(owner)[0] == 0 and len(set(owner)) == 3
    assert lpt_assign(costs, 1) == [0] * 8


def test_dictionary_and_bow():
    from lda_thesis_amd.text import Dictionary
    docs = [["b", "a", "b"], ["c", "a"], ["d"]]
    d = Dictionary(docs)
    assert d.token2id == {"a": 0, "b": 1, "c": 2, "d": 3} and len(d) == 4
    assert d.doc2bow(["b", "zz", "a", "b"]) == [(0, 1), (1, 2)]
    assert d.values() == ["a", "b", "c", "d"]
    d.filter_extremes(no_below=2, no_above=1.0)
    assert d.token2id == {"a": 0} and d.doc2bow(["a", "b"]) == [(0, 1)]


def test_preprocess_and_stemmer():
    from lda_thesis_amd.text import porter_stem, preprocess_documents, preprocess_string, simple_preprocess
    assert simple_preprocess("The 3 Models of <b>Growth</b>, and a tax!") == ["models", "growth", "tax"]
    # the product pipeline: gensim's filter order, stemmed; digits vanish and the letters around them join
    assert preprocess_string("The 3 Models of <b>Growth</b>, and a tax!") == ["model", "growth", "tax"]
    assert preprocess_string("abc123def policies; tax-rates of 1990s") == ["abcdef", "polici", "tax", "rate"]
    assert preprocess_documents(["Taxation and economic growth"]) == [["taxat", "econom", "growth"]]
    # Porter (1980): the examples of the paper's steps carried through the whole algorithm
    pairs = """caresses caress  ponies poni  ties ti  caress caress  cats cat  feed feed  agreed agre  plastered plaster
               bled bled  motoring motor  sing sing  conflated conflat  troubled troubl  sized size  hopping hop
               tanned tan  falling fall  hissing hiss  fizzed fizz  failing fail  filing file  happy happi  sky sky
               relational relat  conditional condit  rational ration  digitizer digit  operator oper
               feudalism feudal  decisiveness decis  hopefulness hope  callousness callous  electrical electr
               hopeful hope  goodness good  revival reviv  allowance allow  inference infer  airliner airlin
               adjustable adjust  defensible defens  irritant irrit  replacement replac  adjustment adjust
               dependent depend  adoption adopt  communism commun  activate activ  effective effect
               bowdlerize bowdler  probate probat  rate rate  cease ceas  controlling control  rolling roll
               generalization gener  generalizations gener  economics econom  economy economi  taxes tax
               oscillators oscil  markets market""".split()
    for w, s in zip(pairs[0::2], pairs[1::2]):
        assert porter_stem(w) == s, (w, porter_stem(w), s)


def test_load_corpus_label_parsing(tmp_path):
    from lda_thesis_amd import CascadeLDA as C, LabeledLDA as L
    p = tmp_path / "c.csv"
    p.write_text('id1,"growth and taxes","E32 H20 xx"\nid2,"labor markets",J\nid3,"empty labels",\n'
                 'id4,"more growth","E32 E31"\n')
    docs, labs, labelset = L.load_corpus(str(p), 2)
    assert [sorted(x) for x in labs] == [["E3", "H2"], ["J"], [""], ["E3"]]
    assert labelset == ["E3", "H2", "J", ""]
    assert docs[0] == ["growth", "tax"]                # stemmed, as gensim's preprocess_documents does
    docs, labs, labelset = C.load_corpus(str(p), 3)
    assert sorted(labs[0]) == ["E", "E3", "E32", "H", "H2", "H20"]
    assert labs[1] == ["J", "J", "J"] and labs[2] == ["", "", ""]
    assert C.partition_label("E32", 3) == ["E", "E3", "E32"]


def test_csr_from_doc_tups():
    from lda_thesis_amd.corpus import csr_from_doc_tups
    off, w, f = csr_from_doc_tups([[(1, 2), (5, 1)], [(0, 3)]])
    assert off.tolist() == [0, 2, 3] and w.tolist() == [1, 5, 0] and f.tolist() == [2, 1, 3]


def test_cascade_enumeration_order_and_subcorpus():
    from fixture_corpora import cascade_corpus
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.text import Dictionary
    docs, labs, labelset = cascade_corpus()
    c = CascadeLDA(docs, labs, list(labelset), Dictionary(docs), 0.1, 0.01)
    tasks = c.enumerate_subproblems()
    parents = [t["parent"] for t in tasks]
    assert parents[0] == "root" and set(parents[1:]) == set(c.lablist_l1 + c.lablist_l2)
    # depth-first: every level-2 parent follows its level-1 parent before the next level-1 label
    for i, p in enumerate(parents[1:], 1):
        if len(p) == 2:
            j = max(k for k in range(i) if len(parents[k]) == 1)
            assert parents[j] == p[0]
    t = tasks[parents.index("A")]
    assert t["labset"] == sorted(t["labset"]) and all(x[0] == "A" and len(x) == 2 for x in t["labset"])
    assert all(any(True for _ in lab) or True for lab in t["labs"]) and len(t["doc_tups"]) == len(t["labs"])


def test_synthetic_corpus_properties():
    import torch
    from lda_thesis_amd.corpus import synthetic_corpus
    off, w, f, z = synthetic_corpus(200, 30, 500, 16, seed=3, device="cpu", chunk=64)
    w = w.view(200, 30)
    assert (w[:, 1:] > w[:, :-1]).all() and int(w.max()) < 500 and int(w.min()) >= 0
    assert (f == 1).all() and off[-1] == 6000 and int(z.max()) < 16
    off2, w2, _, _ = synthetic_corpus(200, 30, 500, 16, seed=3, device="cpu", chunk=64)
    assert torch.equal(w2.view(200, 30), w)
    head = (w < 50).float().mean()
    assert head > 0.3                               # Zipf: the 10% most frequent ids dominate


def test_vectorised_choice_equals_numpy_choice():
    """ensemble.draw_initial_topics restates numpy's legacy choice(K, size, p=lab/lab.sum()) (what the reference
    draws per document, /root/reference/CascadeLDA.py:373-381): same uniforms, same topics, same stream position."""
    from lda_thesis_amd.ensemble import draw_initial_topics
    rng = np.random.default_rng(12)
    for K in (2, 3, 7, 20, 57):
        D = 300
        labs = (rng.random((D, K)) < 0.35).astype(float)
        labs[:, 0] = 1.0
        lens = rng.integers(1, 40, size=D)
        np.random.seed(K)
        want = np.concatenate([np.random.choice(K, size=n, p=lab / lab.sum()) for lab, n in zip(labs, lens)])
        after_want = np.random.random_sample()
        np.random.seed(K)
        u = np.random.random_sample(int(lens.sum()))
        after_got = np.random.random_sample()
        n_allowed = labs.sum(axis=1).astype(np.int64)
        a_max = int(n_allowed.max())
        allowed = np.full((D, a_max), -1, dtype=np.int64)
        for d in range(D):
            ids = np.flatnonzero(labs[d])
            allowed[d, :len(ids)] = ids
        got = draw_initial_topics(allowed, n_allowed, np.repeat(np.arange(D), lens), u)
        np.testing.assert_array_equal(got, want)
        assert after_got == after_want


def test_labeledlda_initial_topics_equal_the_reference_loop():
    """LabeledLDA._initial_topics (one random_sample block + table lookups) == the reference's per-document
    np.random.choice(K, size=len(doc), p=lab / lab.sum()) (/root/reference/LabeledLDA.py:80-88): same topics, same stream position."""
    from lda_thesis_amd.LabeledLDA import LabeledLDA
    rng = np.random.default_rng(5)
    for K, most in ((392, 8), (40, 39), (3, 2)):
        D = 500
        labs = np.zeros((D, K))
        labs[:, 0] = 1.0
        for d in range(D):
            labs[d, rng.choice(K - 1, size=int(rng.integers(0, most + 1)), replace=False) + 1] = 1.0
        lens = rng.integers(1, 60, size=D)
        np.random.seed(K)
        want = np.concatenate([np.random.choice(K, size=n, p=lab / lab.sum()) for lab, n in zip(labs, lens)])
        after_want = np.random.randint(0, 2 ** 31 - 1)
        m = object.__new__(LabeledLDA)
        m.labs, m.D, m.K = labs, D, K
        np.random.seed(K)
        got = m._initial_topics(lens)
        assert np.random.randint(0, 2 ** 31 - 1) == after_want
        np.testing.assert_array_equal(got, want)


def test_plan_subproblems_equals_the_list_based_enumeration():
    """CascadeLDA.plan_subproblems (index arrays for the batched ensemble) describes exactly the sub-problems
    enumerate_subproblems + SubLDA.__init__ build (reference CascadeLDA.py:113-127, 135-184, 347-392), and the
    vectorised initial assignments equal the per-document np.random.choice stream."""
    import sys
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from fixture_corpora import cascade_corpus
    from lda_thesis_amd.CascadeLDA import CascadeLDA, SubLDA
    from lda_thesis_amd.ensemble import draw_initial_topics
    from lda_thesis_amd.text import Dictionary
    docs, labs, labelset = cascade_corpus()
    labs[3] = labs[3] + [labs[3][0]]                         # a duplicated label, as short label fields produce
    dicti = Dictionary(docs)
    np.random.seed(3)
    c = CascadeLDA(docs, labs, list(labelset), dicti, 0.1, 0.01, seed=5)
    plans = c.plan_subproblems()
    tasks = c.enumerate_subproblems()
    assert [p["parent"] for p in plans] == [t["parent"] for t in tasks]
    lens = np.array([len(t) for t in c.doc_tups])
    np.random.seed(99)
    subs = [SubLDA(t["doc_tups"], t["labs"], t["labset"], dicti, alpha=0.1, beta=0.01, seed=5, stream_id=i, defer=True)
            for i, t in enumerate(tasks)]
    np.random.seed(99)
    for pl, t, sub in zip(plans, tasks, subs):
        assert ["root"] + pl["labset"] == t["labset"] and pl["K"] == sub.K      # SubLDA inserted 'root'
        assert [c.doc_tups[d] for d in pl["docs"]] == t["doc_tups"]
        flags = np.zeros((len(pl["docs"]), pl["K"]))
        for r, (row, n) in enumerate(zip(pl["allowed"], pl["n_allowed"])):
            assert (row[:n] >= 0).all() and (row[n:] == -1).all() and (np.diff(row[:n]) > 0).all()
            flags[r, row[:n]] = 1.0
        np.testing.assert_array_equal(flags, sub.labs)
        n_sites = lens[pl["docs"]]
        u = np.random.random_sample(int(n_sites.sum()))
        z = draw_initial_topics(pl["allowed"], pl["n_allowed"], np.repeat(np.arange(len(pl["docs"])), n_sites), u)
        np.testing.assert_array_equal(z, sub._z0)
        t["labset"].remove("root")


@pytest.mark.parametrize("K", [2, 7, 8, 9, 20, 40, 127, 128, 129, 130, 300, 1031])
def test_device_column_normalisation_is_numpys_pairwise_order(K):
    """``ph[:, ids]`` is F-contiguous, so the reference's ``probs.sum(axis=0)`` (LabeledLDA.py:161, CascadeLDA.py:195)
    reduces every column with numpy's pairwise sum -- for K >= 8 not the sequential sum over the rows.  The torch
    statements that prepare the fold-in on the device must give the same bits as the numpy expressions."""
    import torch
    from lda_thesis_amd.foldin import cascade_init_rows, cascade_init_rows_device, numpy_column_sums
    from lda_thesis_amd.layout import group_layout
    rng = np.random.default_rng(K)
    V = 37
    ph = rng.random((K, V)) ** 6
    ph[rng.random((K, V)) < 0.3] = 0.0
    ids = [0, 3, 4, 11, 36, 20, 7]
    np.testing.assert_array_equal(numpy_column_sums(torch.from_numpy(ph)).numpy()[ids], ph[:, ids].sum(axis=0))
    tups = [[(3, 1), (4, 2), (20, 1)], [(0, 5)], [(7, 1), (11, 1), (36, 2), (3, 1)]]
    want, _ = cascade_init_rows(ph.copy(), 0.01, tups)
    lay = group_layout(K)
    doc_off = np.array([0, 3, 4, 8])
    word = torch.tensor([3, 4, 20, 0, 7, 11, 36, 3])
    got = cascade_init_rows_device(torch.from_numpy(ph.copy()), 0.01, doc_off, word, lay)
    np.testing.assert_array_equal(got[:, torch.from_numpy(lay.lm_topic_pos.astype(np.int64))].numpy(), want)


def test_head_many_equals_head_row_by_row():
    """CascadeLDA._head_many (the batched tree walk) selects what _head -- the statements of test_down_tree,
    /root/reference/CascadeLDA.py:253-258 -- selects for every row, ties and exactly uniform rows included."""
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    rng = np.random.default_rng(0)
    for K in (2, 5, 20, 27):
        th = rng.dirichlet(np.ones(K) * 0.3, size=200)
        th[5] = th[6]
        th[7, :] = 1.0 / K
        th[8, :2] = th[8, 0]
        if K == 5:
            th = np.round(th, 3)
        labels = ["L%d" % i for i in range(K)]
        many = CascadeLDA._head_many(th, labels, 0.95)
        assert len(many) == th.shape[0]
        for r in range(th.shape[0]):
            keep, loads = CascadeLDA._head(th[r], labels, 0.95)
            assert keep == many[r][0]
            np.testing.assert_array_equal(loads, many[r][1])
    assert CascadeLDA._head_many(np.zeros((0, 4)), ["a", "b", "c", "d"], 0.9) == []


def test_doc2bow_counts_known_tokens_in_id_order():
    """Dictionary.doc2bow: sorted (id, count) pairs of the in-vocabulary tokens (gensim's contract, which
    /root/reference/LabeledLDA.py:64,156 relies on: ids unique and ascending inside a document)."""
    from lda_thesis_amd.text import Dictionary
    d = Dictionary([["b", "a", "c"], ["c", "d"]])
    ids = d.token2id
    bow = d.doc2bow(["c", "zzz", "a", "c", "c", "d", "a", "nope"])
    assert bow == sorted([(ids["a"], 2), (ids["c"], 3), (ids["d"], 1)])
    assert d.doc2bow([]) == [] and d.doc2bow(["zzz"]) == []
