"""lda_thesis_amd.evaluate and the label parsing of load_corpus against input/output pairs produced by
the reference's evaluate_LabeledLDA.py / evaluate_CascadeLDA.py / load_corpus (oracle/gen_golden.py)."""
import json

import numpy as np
import pytest

from conftest import load_golden


def test_metrics_match_reference():
    from lda_thesis_amd import evaluate as E
    g = load_golden("evaluate")
    th, y = g["th"], g["y"]
    tps, tns, fps, fns, fprs, tprs = E.rates(th, y)
    want = json.loads(str(g["rates"]))
    for got, w in zip((tps, tns, fps, fns, fprs, tprs), want):
        assert [[float(v) for v in doc] for doc in got] == w
    assert E.n_error(th, y, 1) == float(g["n_error1"]) and E.n_error(th, y, 2) == float(g["n_error2"])
    assert E.macro_auc_roc(fprs, tprs) == float(g["auc"])
    assert E.get_f1(tps, fps, tns, fns) == float(g["f1"])


def test_binary_yreal_and_setup_theta():
    from lda_thesis_amd import evaluate as E
    g = load_golden("evaluate")
    l1p, l2p, l3p, labmap = json.loads(str(g["setup_theta_in"]))
    tup = lambda lvl: [[tuple(x) for x in grp] for grp in lvl]
    l1p = [[tuple(x) for x in d] for d in l1p]
    l2p, l3p = [tup(d) for d in l2p], [tup(d) for d in l3p]

    class M(object):
        labelmap = labmap
    np.testing.assert_array_equal(E.setup_theta(l1p, l2p, l3p, M()), g["setup_theta"])
    np.testing.assert_array_equal(E.binary_yreal([["A", "A1", "zz"], [], ["B11", "root"]], labmap), g["binary_yreal"])


def test_rate_without_negatives_is_nan_like_reference():
    from lda_thesis_amd import evaluate as E
    with np.errstate(invalid="ignore"):
        fprs = E.rates(np.array([[0.6, 0.4]]), np.array([[1, 1]]))[4]
    assert np.isnan(fprs[0]).all()


def test_label_parsing_matches_reference_load_corpus(tmp_path):
    from lda_thesis_amd import CascadeLDA as C, LabeledLDA as L
    g = load_golden("evaluate")
    p = tmp_path / "c.csv"
    p.write_text(str(g["csv_text"]))
    want = json.loads(str(g["parsed"]))
    for d in (1, 2, 3):
        _, labs, labelset = L.load_corpus(str(p), d)
        assert [[sorted(x) for x in labs], labelset] == want["llda_%d" % d]
    _, labs, labelset = C.load_corpus(str(p), 3)
    # the reference registers labels while iterating a python set (CascadeLDA.py:35-37), so the order
    # inside one row depends on the process' string-hash seed: compare per-row label sets and the
    # order in which ROWS introduce new labels
    assert [sorted(set(x)) for x in labs] == want["cascade_3"][0]
    assert sorted(labelset) == sorted(want["cascade_3"][1]) and len(labelset) == len(set(labelset))

    def first_row(ls, rows):
        return [min(i for i, r in enumerate(rows) if x in r) for x in ls]
    fr = first_row(labelset, labs)
    assert fr == sorted(fr)
