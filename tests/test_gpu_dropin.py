"""The drop-in classes (lda_thesis_amd.LabeledLDA / CascadeLDA) driven through the reference's own
constructor and method names, against golden vectors produced by the reference classes themselves."""
import pickle

import numpy as np
import pytest

from conftest import load_golden
from helpers import assert_state_equal

pytestmark = pytest.mark.gpu


def build_model(name, seed=12345):
    from fixture_corpora import tiny_corpus
    from lda_thesis_amd.LabeledLDA import LabeledLDA
    from lda_thesis_amd.text import Dictionary
    docs, labs, labelset, alpha, beta, sweeps, npseed = tiny_corpus(name)
    dicti = Dictionary(docs)
    ls = list(labelset)
    np.random.seed(npseed)
    m = LabeledLDA(docs, labs, ls, dicti, alpha, beta, seed=seed)
    return m, ls, sweeps


@pytest.mark.parametrize("name", ["k05", "k12", "k20dense", "k40", "k130", "k392", "k1031", "k2100"])
def test_labeledlda_init_and_sweeps(name):
    g = load_golden("tiny_" + name)
    m, ls, sweeps = build_model(name)
    assert ls[0] == "root" and m.labelmap["root"] == 0 and m.K == int(g["K"]) and m.V == int(g["V"])
    # same np.random.choice stream as the reference constructor => identical initial state
    assert_state_equal(g, "init", m.n_k_v, m.n_d_k, m.n_zk, np.concatenate(m.z_dn), name + " init")
    assert m.n_k_v.dtype == np.int64 and m.n_k_v.shape == (m.K, m.V) and m.n_d_k.shape == (m.D, m.K)
    for i in range(sweeps):
        m.training_iteration()
        assert_state_equal(g, "o3_s%d" % (i + 1), m.n_k_v, m.n_d_k, m.n_zk, np.concatenate(m.z_dn), name)
    np.testing.assert_array_equal(m.get_phi(), g["o3_phi"])
    np.testing.assert_array_equal(m.get_theta(), g["o3_theta"])
    assert abs(m.perplexity() / float(g["o3_perplexity"]) - 1) < 1e-9        # bar: 1e-5 relative


def test_run_training_matches_reference(capsys):
    g = load_golden("runtraining_k12")
    m, _, _ = build_model("k12", seed=int(g["seed"]))
    m.run_training(int(g["iters"]), int(g["thinning"]))
    out = capsys.readouterr().out
    assert out.count("Running iteration # ") == int(g["iters"])
    np.testing.assert_array_equal(m.ph_hat, g["ph_hat"])
    np.testing.assert_array_equal(m.th_hat, g["th_hat"])
    np.testing.assert_allclose(np.array(m.cur_perplx), g["cur_perplx"], rtol=1e-9)


def test_pickle_roundtrip_continues_identically():
    g = load_golden("tiny_k12")
    m, _, sweeps = build_model("k12")
    m.training_iteration()
    m2 = pickle.loads(pickle.dumps(m))
    for i in range(1, sweeps):
        m2.training_iteration()
    assert_state_equal(g, "o3_s%d" % sweeps, m2.n_k_v, m2.n_d_k, m2.n_zk, np.concatenate(m2.z_dn))


def test_empty_document_raises_like_reference():
    from lda_thesis_amd.LabeledLDA import LabeledLDA
    from lda_thesis_amd.text import Dictionary
    docs = [["aa", "bb"], ["zz"]]
    dicti = Dictionary([docs[0]])                       # 'zz' is out of vocabulary
    with pytest.raises(ValueError):
        LabeledLDA(docs, [["x"], ["x"]], ["x"], dicti, 0.1, 0.01)


def test_sublda_phantom_init_and_run_training(capsys):
    from fixture_corpora import tiny_corpus
    from lda_thesis_amd.CascadeLDA import SubLDA
    from lda_thesis_amd.text import Dictionary
    g = load_golden("runtraining_k12")
    docs, labs, labelset, alpha, beta, _, npseed = tiny_corpus("k12")
    dicti = Dictionary(docs)
    doc_tups = [dicti.doc2bow(x) for x in docs]
    np.random.seed(npseed + 1)
    sub = SubLDA(doc_tups, labs, list(labelset), dicti, alpha=alpha, beta=beta, seed=int(g["seed"]),
                 stream_id=int(g["sub_stream"]))
    assert sub.n_k_v.sum() > sub.n_zk.sum()             # phantom columns (CascadeLDA.py:382-385)
    sub.run_training(it=int(g["sub_it"]), thinning=int(g["sub_thinning"]))
    assert "Training iteration # 2" in capsys.readouterr().out
    np.testing.assert_array_equal(sub.ph, g["sub_ph"])
    np.testing.assert_array_equal(sub.get_ph(), g["sub_get_ph"])


def test_sublda_golden_phantom_counts():
    """phantom initialisation == the reference's n_k_v for the sublda fixture: the host statement and the device
    path (llda_count_init + add_word_topic_counts) SubLDA actually uses."""
    from lda_thesis_amd.CascadeLDA import SubLDA
    g = load_golden("sublda")
    sub = SubLDA.__new__(SubLDA)
    off = g["doc_off"]
    sub.doc_tups = [list(zip(g["word"][off[d]:off[d + 1]].tolist(), g["freq"][off[d]:off[d + 1]].tolist()))
                    for d in range(int(g["D"]))]
    sub._z0, sub.D, sub.K, sub.V = g["init_z"].astype(np.int64), int(g["D"]), int(g["K"]), int(g["V"])
    _, counts = sub._initial_counts()
    np.testing.assert_array_equal(counts["n_k_v"], g["init_n_k_v"])
    np.testing.assert_array_equal(counts["n_d_k"], g["init_n_d_k"])
    np.testing.assert_array_equal(counts["n_zk"], g["init_n_zk"])
    sub.labs, sub.alpha, sub.beta = g["labs"].astype(float), float(g["alpha"]), float(g["beta"])
    sub.seed, sub.stream_id, sub._device, sub._sampler = 1, 0, None, None
    np.testing.assert_array_equal(sub.n_k_v, g["init_n_k_v"])
    np.testing.assert_array_equal(sub.n_d_k, g["init_n_d_k"])
    np.testing.assert_array_equal(sub.n_zk, g["init_n_zk"])


def test_cascade_go_down_tree_matches_reference(capsys):
    from fixture_corpora import cascade_corpus
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.text import Dictionary
    g = load_golden("cascade_toy")
    docs, labs, labelset = cascade_corpus()
    dicti = Dictionary(docs)
    np.random.seed(int(g["np_seed"]))
    c = CascadeLDA(docs, labs, list(labelset), dicti, float(g["alpha"]), float(g["beta"]), seed=int(g["seed"]))
    assert list(c.labelmap.keys()) == [str(x) for x in g["labelset"]]
    c.go_down_tree(it=int(g["it"]), s=int(g["s"]))
    assert "root" not in c.lablist_l1                   # inserted by SubLDA, removed again (quirk 1)
    np.testing.assert_array_equal(c.ph, g["ph"])
    out = capsys.readouterr().out
    assert out.count("Working on parent node") == len(g["sizes"]) - 1


def test_cascade_batched_ensemble_equals_one_by_one():
    """the batched ensemble (llda_sweep_batch: all sub-problems in one launch) leaves the state the sub-problems
    reach one by one through llda_sweep, sub-problem by sub-problem -- also when sites are forced through the exact
    pipeline inside the batched kernel."""
    import llda_oracle as orc
    from fixture_corpora import cascade_corpus
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.text import Dictionary
    g = load_golden("cascade_toy")
    docs, labs, labelset = cascade_corpus()
    dicti = Dictionary(docs)

    def run(**kw):
        np.random.seed(int(g["np_seed"]))
        c = CascadeLDA(docs, labs, list(labelset), dicti, float(g["alpha"]), float(g["beta"]), seed=int(g["seed"]))
        for k, v in kw.items():
            setattr(c, k, v)
        return c
    a = run()
    a.go_down_tree(4, 2, batched=True, keep_state=True)
    assert a._ensemble is not None and int(a._ensemble.status[2]) == 0      # production margin: no exact-tier site
    after_a = np.random.random_sample()
    b = run(_keep_subs=[])
    b.go_down_tree(4, 2, batched=False)
    after_b = np.random.random_sample()
    assert after_a == after_b                                  # both consumed numpy's stream identically
    np.testing.assert_array_equal(a.ph, g["ph"])
    np.testing.assert_array_equal(b.ph, g["ph"])
    assert len(b._keep_subs) == len(g["sizes"])
    for i, sub in b._keep_subs:
        n_k_v, n_d_k, n_zk, z = a._ensemble.problem_state(i)
        assert orc.digest(n_k_v, n_d_k, n_zk, z) == orc.digest(sub.n_k_v, sub.n_d_k, sub.n_zk, np.concatenate(sub.z_dn))
    # every site through the exact pipeline INSIDE the batched kernel (exact_site_wave), and a widened margin that
    # mixes decided and exact sites inside one wavefront: same state
    for margin in (-1, 4):
        c = run(_batch_debug_margin=margin)
        c.go_down_tree(4, 2, batched=True, keep_state=True)
        assert c._ensemble is not None and int(c._ensemble.status[2]) > 0
        np.testing.assert_array_equal(c.ph, g["ph"])
        assert np.random.random_sample() == after_a


def test_cascade_tiny_priors_take_the_one_by_one_path():
    """priors below 1e-6 are outside what the batched (sparse) arithmetic proves; go_down_tree then trains the
    sub-problems through llda_sweep's general kernel -- and both settings agree on a prior both can run."""
    from fixture_corpora import cascade_corpus
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.text import Dictionary
    docs, labs, labelset = cascade_corpus()
    dicti = Dictionary(docs)
    np.random.seed(1)
    c = CascadeLDA(docs, labs, list(labelset), dicti, 1e-9, 0.01, seed=3)
    c.go_down_tree(2, 1, keep_state=True)
    assert c._ensemble is None and np.isfinite(c.ph[1:]).any()


def test_cascade_abstracts_ensemble_matches_reference():
    """BASELINE configs[4] at its real size: go_down_tree(4, 2) on the abstracts corpus, 122 sub-problems, against
    the reference's own go_down_tree run with per-document-snapshot sweeps (oracle/gen_golden.py
    gen_cascade_abstracts; /root/reference/CascadeLDA.py:113-184, 347-434): every sub-problem's integer state after
    its last sweep, and the final ph, bit for bit."""
    import hashlib
    import llda_oracle as orc
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.corpus import cascade_corpus_from_csr
    from lda_thesis_amd.text import Dictionary
    g = load_golden("cascade_abstracts")
    a = load_golden("abstracts_d3")
    names = [str(x) for x in a["labelset"]]
    docs, labs, labelset = cascade_corpus_from_csr(a["doc_off"], a["word"], a["freq"], a["lab_off"], a["lab_idx"], names)
    dicti = Dictionary(docs)
    np.random.seed(int(g["np_seed"]))
    c = CascadeLDA(docs, labs, list(labelset), dicti, float(g["alpha"]), float(g["beta"]), seed=int(g["seed"]))
    assert list(c.labelmap.keys()) == [str(x) for x in g["labelset"]]
    c.go_down_tree(int(g["it"]), int(g["s"]), keep_state=True)
    ens = c._ensemble
    assert ens is not None and len(ens.plans) == len(g["sizes"]) == 122
    for i in range(len(ens.plans)):
        n_k_v, n_d_k, n_zk, z = ens.problem_state(i)
        assert (n_d_k.shape[0], n_d_k.shape[1], z.shape[0]) == tuple(int(x) for x in g["sizes"][i]), i
        assert orc.digest(n_k_v, n_d_k, n_zk, z) == str(g["digests"][i]), "sub-problem %d" % i
    np.testing.assert_array_equal(c.ph[:, ::53], g["ph_sample"])
    ph = np.ascontiguousarray(c.ph)
    with np.errstate(invalid="ignore"):
        ph[np.isnan(ph)] = (np.zeros(1) / np.zeros(1))[0]      # the bit pattern numpy's own 0/0 has (a never-used child)
    assert hashlib.sha256(ph.tobytes()).hexdigest() == str(g["ph_sha256"])


@pytest.mark.parametrize("name", ["k12", "k40", "k130", "k1031", "k2100"])
def test_fold_in_matches_reference_run_test(name):
    """llda_foldin (prep4test + run_test on the device) vs the reference's run_test with the keyed draw."""
    from lda_thesis_amd.foldin import fold_in
    g = load_golden("runtest_" + name)
    off = g["doc_off"]
    tups = [list(zip(g["word"][off[d]:off[d + 1]].tolist(), g["freq"][off[d]:off[d + 1]].tolist()))
            for d in range(len(off) - 1)]
    r = fold_in(g["ph_hat"], float(g["alpha"]), tups, int(g["it"]), int(g["thinning"]), int(g["seed"]),
                stream_id=int(g["stream"]))
    np.testing.assert_array_equal(r["th_hat"], g["th_hat"])
    assert (r["n_dk"].sum(1) == [sum(f for _, f in t) for t in tups]).all()


def test_fold_in_against_numpy_oracle_on_seeded_inputs():
    """bigger K / longer documents / more sweeps than the committed golden, vs oracle/llda_oracle.run_test."""
    import llda_oracle as orc
    from lda_thesis_amd.foldin import fold_in
    rng = np.random.default_rng(5)
    for K, V in ((7, 40), (64, 90), (200, 60), (512, 50), (1100, 30), (3003, 24)):     # the last two: wide layouts
        ph = rng.random((K, V)) ** 3
        ph[rng.random((K, V)) < 0.2] = 0.0
        ph /= ph.sum(axis=1, keepdims=True)
        docs, freqs, tups = [], [], []
        for d in range(12):
            ids = np.sort(rng.choice(V, size=int(rng.integers(1, 25)), replace=False)).tolist()
            fr = rng.integers(1, 4, size=len(ids)).tolist()
            docs.append(ids); freqs.append(fr); tups.append(list(zip(ids, fr)))
        want = orc.run_test(ph, 0.3, docs, freqs, 7, 3, orc.keyed_draw_for(99, 4, doc_base=10))
        got = fold_in(ph, 0.3, tups, 7, 3, 99, stream_id=4, doc_base=10)
        np.testing.assert_array_equal(got["th_hat"], want)


@pytest.mark.parametrize("K,V,dense", [(5, 60, True), (20, 80, False), (64, 90, True), (392, 70, False), (512, 50, True),
                                       (1000, 40, False)])
def test_fold_in_decided_tier_equals_the_reference_pipeline(K, V, dense):
    """The fold-in kernel decides a site from unnormalised fp64 prefix sums when every |prefix - u * total| exceeds 2^-40 of
    the total (DESIGN.md 4.3 applied to LabeledLDA.py:183-197 / CascadeLDA.py:216-236) and runs the reference's pipeline
    (normalise, shrink, inverse-CDF draw) otherwise; ``exact_only`` forces the latter for every site.  Same z, n_dk, th_hat,
    bit for bit, on loadings with a wide dynamic range, exact zeros, all-zero columns (the beta fall-back) and tiny alpha."""
    import lda_thesis_amd.foldin as F
    rng = np.random.default_rng(K + V)
    ph = rng.random((K, V)) ** 12                                   # 12 decades between the loadings of a word
    if not dense:
        ph[rng.random((K, V)) < 0.5] = 0.0
        ph[:, 3] = 0.0                                              # a word no topic loads on
        ph[0, :] += 1e-9                                            # (no all-zero row)
    ph /= ph.sum(axis=1, keepdims=True)
    tups = []
    for d in range(40):
        ids = np.sort(rng.choice(V, size=int(rng.integers(1, 30)), replace=False)).tolist()
        tups.append(list(zip(ids, rng.integers(1, 5, size=len(ids)).tolist())))
    runs = {}
    for mode in (False, True):
        F.EXACT_ONLY = mode
        try:
            a = F.cascade_fold_in(ph, 1e-3, 0.01, tups, 40, 5, 77, 4242, np.arange(len(tups)) + 3)
            b = None if not dense else F.fold_in(ph, 0.2, tups, 40, 5, 78, stream_id=9, doc_base=2)
        finally:
            F.EXACT_ONLY = False
        runs[mode] = (a, b)
    for got, want in zip(runs[False], runs[True]):
        if got is None:
            continue
        np.testing.assert_array_equal(got["th_hat"], want["th_hat"])
        np.testing.assert_array_equal(got["n_dk"], want["n_dk"])
        for zg, zw in zip(got["z"], want["z"]):
            np.testing.assert_array_equal(zg, zw)


def test_labeledlda_run_test_method():
    g = load_golden("runtest_k12")
    m, _, _ = build_model("k12", seed=12345)
    m.run_training(4, 2)
    np.testing.assert_array_equal(m.ph_hat, g["ph_hat"])            # same trained model as the fixture
    off = g["doc_off"]
    inv = {v: k for k, v in m.dicti.token2id.items()}
    newdocs = [[inv[w] for w, f in zip(g["word"][off[d]:off[d + 1]], g["freq"][off[d]:off[d + 1]]) for _ in range(f)]
               for d in range(len(off) - 1)]
    th = m.run_test(newdocs, int(g["it"]), int(g["thinning"]), seed=int(g["seed"]), stream_id=int(g["stream"]))
    np.testing.assert_array_equal(th, g["th_hat"])
    ids, freqs, z_dn, n_dk = m.prep4test(newdocs[0])
    assert len(z_dn) == len(ids) and n_dk.sum() == sum(freqs)


def cascade_model():
    from fixture_corpora import cascade_corpus
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.text import Dictionary
    g = load_golden("cascade_toy")
    docs, labs, labelset = cascade_corpus()
    dicti = Dictionary(docs)
    np.random.seed(int(g["np_seed"]))
    c = CascadeLDA(docs, labs, list(labelset), dicti, float(g["alpha"]), float(g["beta"]), seed=4321)
    c.ph = g["ph"].copy()
    return c, dicti


def held_out_docs(g, dicti):
    inv = {v: k for k, v in dicti.token2id.items()}
    off = g["doc_off"]
    return [[inv[w] for w, f in zip(g["word"][off[d]:off[d + 1]], g["freq"][off[d]:off[d + 1]]) for _ in range(f)]
            for d in range(len(off) - 1)]


def test_cascade_test_down_tree_matches_reference():
    import json
    g = load_golden("cascade_test_toy")
    c, dicti = cascade_model()
    docs = held_out_docs(g, dicti)
    want = json.loads(str(g["trees"]))
    for doc, w in zip(docs, want):
        l1, l2, l3 = c.test_down_tree(doc, int(g["it"]), int(g["thinning"]), float(g["threshold"]))
        assert [(a, float(b)) for a, b in l1] == [tuple(x) for x in w[0]]
        assert [[(a, float(b)) for a, b in grp] for grp in l2] == [[tuple(x) for x in grp] for grp in w[1]]
        assert [[(a, float(b)) for a, b in grp] for grp in l3] == [[tuple(x) for x in grp] for grp in w[2]]
    np.testing.assert_array_equal(c.cascade_test(docs[0], 4, 1, ["A", "A1", "A2"]), g["single_A"])
    # the batch form (one launch per node of the label tree, all documents that reach it) returns the same trees
    got = c.test_down_tree_batch(docs, int(g["it"]), int(g["thinning"]), float(g["threshold"]))
    assert len(got) == len(docs)
    for (l1, l2, l3), w in zip(got, want):
        assert [(a, float(b)) for a, b in l1] == [tuple(x) for x in w[0]]
        assert [[(a, float(b)) for a, b in grp] for grp in l2] == [[tuple(x) for x in grp] for grp in w[1]]
        assert [[(a, float(b)) for a, b in grp] for grp in l3] == [[tuple(x) for x in grp] for grp in w[2]]


def test_cascade_flat_run_test_matches_reference(capsys):
    g = load_golden("cascade_test_toy")
    c, dicti = cascade_model()
    docs = held_out_docs(g, dicti)
    np.testing.assert_array_equal(c.run_test(docs, 4, 2, depth="all"), g["flat_all"])
    assert capsys.readouterr().out.count("Testing iteration #") == 2 * len(docs)
    np.testing.assert_array_equal(c.run_test(docs, 4, 2, depth=1), g["flat_1"])
    assert "flat_2_raises" in g                   # the reference raises (NaN pvals) for this label subset ...
    with pytest.raises(ValueError):               # ... and so does the device path
        c.run_test(docs, 4, 2, depth=2)
    with pytest.raises(UnboundLocalError):        # it < thinning: `th` is never bound in the reference
        c.run_test(docs, 1, 2)


def test_cascade_fold_in_against_numpy_oracle():
    """cascade_test / flat run_test restated in the oracle, on seeded inputs with many zero loadings
    (exercises the beta fallback and long `while prob.sum() > 1` loops)."""
    import llda_oracle as orc
    from lda_thesis_amd.foldin import cascade_fold_in
    rng = np.random.default_rng(11)
    for K, V in ((2, 30), (9, 40), (40, 50)):
        ph = rng.random((K, V))
        ph[rng.random((K, V)) < 0.6] = 0.0
        ph[0] += 1e-3
        ph /= ph.sum(axis=1, keepdims=True)
        tups = []
        for d in range(4):
            ids = np.sort(rng.choice(V, size=int(rng.integers(1, 14)), replace=False)).tolist()
            tups.append(list(zip(ids, rng.integers(1, 4, size=len(ids)).tolist())))
        got = cascade_fold_in(ph, 0.2, 0.01, tups, 5, 2, 77, 12345, np.arange(4) + 3)
        for d, tup in enumerate(tups):
            ids, fr = zip(*tup)
            def draw_for_sweep(sw, d=d):
                k = orc.KeyedDraw(77, 12345)
                k.sweep, k.doc, k.site = sw, d + 3, 0
                return k
            want = orc.cascade_test(ph, 0.2, 0.01, list(ids), list(fr), 5, 2, draw_for_sweep)
            np.testing.assert_array_equal(got["th_hat"][d], want, err_msg="K=%d doc %d" % (K, d))
        flat = cascade_fold_in(ph, 0.2, 0.01, tups, 4, 2, 77, 999, np.arange(4), flat=True)
        want = orc.cascade_run_test(ph, 0.2, 0.01, [[v for v, _ in t] for t in tups], [[f for _, f in t] for t in tups],
                                    4, 2, orc.keyed_draw_for(77, 999))
        np.testing.assert_array_equal(flat["th_hat"], want)


def test_cascade_fold_in_wide_layout():
    """the same two samplers on a label subset with more than 8 pairwise leaves (wide fold-in kernels: beta fallback,
    both averaging formulas, per-document RNG ids)."""
    import llda_oracle as orc
    from lda_thesis_amd.foldin import cascade_fold_in
    rng = np.random.default_rng(12)
    K, V = 1031, 16
    ph = rng.random((K, V))
    ph[rng.random((K, V)) < 0.7] = 0.0
    ph[:, 0] += 1e-3                                # (no all-zero row)
    ph[:, 3] = 0.0                                  # a word no topic loads on: the beta fallback
    ph /= ph.sum(axis=1, keepdims=True)
    tups = [[(1, 2), (3, 1), (7, 1)], [(3, 2)], [(0, 1), (2, 1), (3, 1), (9, 3), (15, 1)]]
    got = cascade_fold_in(ph, 0.2, 0.01, tups, 3, 1, 77, 4242, np.arange(3) + 5)
    for d, tup in enumerate(tups):
        ids, fr = zip(*tup)
        def draw_for_sweep(sw, d=d):
            k = orc.KeyedDraw(77, 4242)
            k.sweep, k.doc, k.site = sw, d + 5, 0
            return k
        want = orc.cascade_test(ph, 0.2, 0.01, list(ids), list(fr), 3, 1, draw_for_sweep)
        np.testing.assert_array_equal(got["th_hat"][d], want, err_msg="doc %d" % d)
    tups = [[(v, f) for v, f in t if v != 3] for t in tups if t != [(3, 2)]]     # (run_test has no fallback: 0/0 raises)
    flat = cascade_fold_in(ph, 0.2, 0.01, tups, 4, 2, 77, 999, np.arange(2), flat=True)
    want = orc.cascade_run_test(ph, 0.2, 0.01, [[v for v, _ in t] for t in tups], [[f for _, f in t] for t in tups],
                                4, 2, orc.keyed_draw_for(77, 999))
    np.testing.assert_array_equal(flat["th_hat"], want)


def _write_csv(path, n=120, seed=3):
    rng = np.random.default_rng(seed)
    codes = ["A11", "A12", "A21", "B11", "B21", "B22", "C31"]
    words = ["growth", "taxes", "labor", "market", "policy", "trade", "capital", "wages", "prices", "credit",
             "banking", "income", "health", "energy", "education", "housing", "export", "budget", "inflation"]
    topic_words = {c: rng.choice(len(words), size=6, replace=False) for c in codes}
    lines = []
    for i in range(n):
        labs = [codes[j] for j in rng.choice(len(codes), size=int(rng.integers(1, 3)), replace=False)]
        toks = [words[int(rng.choice(topic_words[labs[int(rng.integers(len(labs)))]]))] for _ in range(int(rng.integers(12, 40)))]
        lines.append('d%d,"%s","%s"' % (i, " ".join(toks), " ".join(labs)))
    path.write_text("\n".join(lines) + "\n")


def test_cli_harness_labeled_lda_end_to_end(tmp_path, capsys, monkeypatch):
    from lda_thesis_amd import evaluate_LabeledLDA as H
    monkeypatch.chdir(tmp_path)
    _write_csv(tmp_path / "toy.csv")
    np.random.seed(0)
    H.main(["-f", str(tmp_path / "toy.csv"), "-d", "3", "-i", "20", "-s", "5", "-p"])
    out = capsys.readouterr().out
    assert "Model:               Labeled LDA" in out and "AUC ROC:" in out and out.count("Running iteration #") == 20
    auc = float(out.split("AUC ROC:")[1].split()[0])
    assert auc > 0.7                                   # topics are recoverable in this toy corpus
    assert (tmp_path / "LabeledLDA_model.pkl").exists()


def test_cli_harness_cascade_end_to_end(tmp_path, capsys, monkeypatch):
    from lda_thesis_amd import evaluate_CascadeLDA as H
    monkeypatch.chdir(tmp_path)
    _write_csv(tmp_path / "toy.csv")
    np.random.seed(1)
    H.main(["-f", str(tmp_path / "toy.csv"), "-i", "8", "-s", "2", "-d", "3"])
    out = capsys.readouterr().out
    assert out.count("Model:               CascadeLDA") == 3 and out.count("AUC ROC:") == 3


def test_integration_md_ctypes_stub_runs_and_matches_reference():
    """the ctypes stub printed in INTEGRATION.md (what a maintainer of the reference would add) is executed
    verbatim against a reference-shaped model object and must reproduce the O3 golden."""
    import os
    import re
    import types
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n# llda_gpu.py.*?```", text, flags=re.S).group(0)
    code = code[len("```python\n"):-3].replace('ctypes.CDLL("lda_thesis_amd/libllda_gibbs.so")',
                                               'ctypes.CDLL(%r)' % os.path.join(ROOT, "lda_thesis_amd", "libllda_gibbs.so"))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = load_golden("tiny_k130")
    off = g["doc_off"]
    D = int(g["D"])
    model = types.SimpleNamespace(
        K=int(g["K"]), V=int(g["V"]), D=D, alpha=float(g["alpha"]), beta=float(g["beta"]),
        docs=[g["word"][off[d]:off[d + 1]].tolist() for d in range(D)],
        freqs=[g["freq"][off[d]:off[d + 1]].tolist() for d in range(D)],
        z_dn=[g["init_z"][off[d]:off[d + 1]].astype(np.int64) for d in range(D)],
        labs=g["labs"].astype(np.float64), n_d_k=g["init_n_d_k"].astype(np.int64),
        n_k_v=g["init_n_k_v"].astype(np.int64), n_zk=g["init_n_zk"].astype(np.int64))
    ns["attach"](model, seed=int(g["seed"]))
    for i in range(int(g["sweeps"])):
        ns["training_iteration"](model)
        ns["pull"](model)
        assert_state_equal(g, "o3_s%d" % (i + 1), model.n_k_v, model.n_d_k, model.n_zk, np.concatenate(model.z_dn))


def test_sampler_validates_frequencies_and_word_ids():
    """inputs the int32 / 24-bit device arithmetic cannot hold are refused on the host, not miscounted."""
    from lda_thesis_amd.sampler import GibbsSampler
    off, w, z = np.array([0, 2]), np.array([0, 1]), np.array([0, 1])
    with pytest.raises(ValueError):
        GibbsSampler(off, w, np.array([1, 1 << 23]), z, 2, 3, 0.1, 0.01)
    with pytest.raises(ValueError):
        GibbsSampler(off, w, np.array([1, -1]), z, 2, 3, 0.1, 0.01)
    with pytest.raises(IndexError):
        GibbsSampler(off, np.array([0, 3]), np.array([1, 1]), z, 2, 3, 0.1, 0.01)
    s = GibbsSampler(off, w, np.array([1, (1 << 23) - 1]), z, 2, 3, 0.1, 0.01)
    s.sweep()
    assert int(s.n_k.sum()) == 1 << 23


def test_training_iteration_loop_raises_where_the_reference_raises():
    """alpha = 0 and a document of ONE site: once the site has left its topic every allowed topic of the document has n_dk + alpha = 0, the
    probabilities are all zero, `prob /= np.sum(prob)` is NaN and numpy's multinomial raises ValueError inside training_iteration
    (/root/reference/LabeledLDA.py:117-119).  A caller that loops training_iteration() -- never run_training -- gets the ValueError
    from the status word: a few sweeps late through the asynchronous copy, at the latest when it reads the counts."""
    from lda_thesis_amd.LabeledLDA import LabeledLDA
    from lda_thesis_amd.text import Dictionary
    docs = [["aa", "bb", "cc", "aa"], ["bb"], ["cc", "dd", "aa"]]
    labs = [["x"], ["y"], ["x", "y"]]
    np.random.seed(3)
    m = LabeledLDA(docs, labs, ["x", "y"], Dictionary(docs), 0.0, 0.01, seed=5)
    with pytest.raises(ValueError, match="positive probability"):
        for _ in range(64):                               # the copy of the status word is polled without synchronising: it lands
            m.training_iteration()                        # within a few sweeps
    m = LabeledLDA(docs, labs, ["x", "y"], Dictionary(docs), 0.0, 0.01, seed=5)
    m.training_iteration()
    with pytest.raises(ValueError, match="positive probability"):
        m.n_d_k                                           # (materialising the counts looks at the status word first)
    # the same corpus with alpha > 0 is fine
    m = LabeledLDA(docs, labs, ["x", "y"], Dictionary(docs), 0.1, 0.01, seed=5)
    for _ in range(8):
        m.training_iteration()
    assert m.n_d_k.sum() == 8
