"""Evaluation metrics of the reference's harness scripts, vectorised.

Mirrors /root/reference/evaluate_LabeledLDA.py:8-107 (one_roc, fpr_tpr, precision_recall, rates,
macro_auc_roc, n_error, get_f1, binary_yreal) and /root/reference/evaluate_CascadeLDA.py:95-127
(setup_theta).  Same names, same return shapes (python lists of per-threshold counts), same corner
cases (the counts are numpy integers as in the reference, so a rate with an empty denominator is
nan/inf with a RuntimeWarning rather than an exception, and get_f1 skips nan with nanmax).  Pinned by tests/golden/evaluate.npz.
"""
import numpy as np


def one_roc(prob, real_binary):
    """confusion counts at every distinct score used as threshold, highest first:
    evaluate_LabeledLDA.py:8-32.  Returns (tp, tn, fp, fn) lists."""
    prob = np.asarray(prob)
    real = np.asarray(real_binary).astype(bool)
    thresholds = np.sort(np.unique(prob))[::-1]
    pred = prob[None, :] >= thresholds[:, None]                  # (thresholds, labels)
    tp = (pred & real).sum(axis=1)
    fp = (pred & ~real).sum(axis=1)
    fn = (~pred & real).sum(axis=1)
    tn = (~pred & ~real).sum(axis=1)
    # numpy integers, as in the reference (sums of numpy booleans): 0/0 below gives nan, not an exception
    return list(tp), list(tn), list(fp), list(fn)


def fpr_tpr(tp, fp, tn, fn):
    fpr = [x / (x + y) for (x, y) in zip(fp, tn)]
    tpr = [x / (x + y) for (x, y) in zip(tp, fn)]
    return fpr, tpr


def precision_recall(tp, fp, tn, fn):
    precis = [x / (x + y) for (x, y) in zip(tp, fp)]
    recall = [x / (x + y) for (x, y) in zip(tp, fn)]
    return precis, recall


def rates(y_prob, y_real_binary):
    """per-document ROC ingredients: evaluate_LabeledLDA.py:47-64."""
    tps, tns, fps, fns, fprs, tprs = [], [], [], [], [], []
    for d_prob, d_real in zip(y_prob, y_real_binary):
        tp, tn, fp, fn = one_roc(d_prob, d_real)
        fpr, tpr = fpr_tpr(tp, fp, tn, fn)
        tps.append(tp); tns.append(tn); fps.append(fp); fns.append(fn)
        fprs.append(fpr); tprs.append(tpr)
    return tps, tns, fps, fns, fprs, tprs


def _auc(x, y):
    """area under a curve by the trapezoidal rule (what sklearn.metrics.auc computes for the
    monotone fpr sequences produced by rates())."""
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    if x.shape[0] < 2:
        raise ValueError("At least 2 points are needed to compute area under curve, but x.shape = %s" % (x.shape,))
    dx = np.diff(x)
    direction = 1
    if np.any(dx < 0):
        if np.all(dx <= 0):
            direction = -1
        else:
            raise ValueError("x is neither increasing nor decreasing : {}.".format(x))
    return direction * float(np.trapezoid(y, x) if hasattr(np, "trapezoid") else np.trapz(y, x))


def macro_auc_roc(fprs, tprs):
    return np.mean([_auc(fpr, tpr) for (fpr, tpr) in zip(fprs, tprs)])


def n_error(th_hat, y_real_binary, n):
    """share of documents with at least one true label among the n highest loads:
    evaluate_LabeledLDA.py:72-82."""
    th_hat = np.asarray(th_hat)
    y = np.asarray(y_real_binary)
    top = np.argsort(th_hat, axis=1)[:, ::-1][:, :n]
    hits = np.take_along_axis(y, top, axis=1).sum(axis=1) > 0
    return int(hits.sum()) / th_hat.shape[0]


def get_f1(tps, fps, tns, fns):
    """mean over documents of the best F1 over the thresholds of the document: evaluate_LabeledLDA.py:85-93.  One array expression
    per document; a threshold without positives on either side gives nan (0 / 0 on numpy integers, as there) and is skipped."""
    best = np.empty(len(tps))
    with np.errstate(invalid="ignore", divide="ignore"):
        for d, (tp, fp, fn) in enumerate(zip(tps, fps, fns)):
            tp, fp, fn = np.asarray(tp), np.asarray(fp), np.asarray(fn)
            prec, rec = tp / (tp + fp), tp / (tp + fn)
            best[d] = np.nanmax((2 * prec * rec) / (prec + rec))
    return np.mean(best)


def binary_yreal(label_strings, label_dict):
    y_true = np.zeros((len(label_strings), len(label_dict)), dtype=int)
    for d, lab in enumerate(label_strings):
        for l in lab:
            ind = label_dict.get(l)
            if ind is not None:
                y_true[d, ind] = 1
    return y_true


def _codes_below(code, codes):
    """the codes that extend ``code`` by exactly one digit, as they occur in ``codes`` (a code can occur inside a longer one: 'C6' is
    found in 'C6' but not in 'C61', whose next character is a digit as well) -- what evaluate_CascadeLDA.py:113-114 asks its
    pattern for."""
    n, out = len(code), []
    for c in codes:
        i = c.find(code)
        while i >= 0:
            j = i + n
            if j < len(c) and c[j].isdigit() and (j + 1 == len(c) or not c[j + 1].isdigit()):
                out.append(c[i:j + 1])
            i = c.find(code, i + 1)
    return out


def setup_theta(l1p, l2p, l3p, model):
    """Fold the per-level predictions of CascadeLDA.test_down_tree into one (documents, labels) matrix: a load is local to its
    parent's sub-problem, so a second-level code is scaled by its first-level code and every third-level code below it by the
    product (evaluate_CascadeLDA.py:95-127).  Per document ONE table code -> load, filled deepest level first (an upper level
    overwrites what a lower one said about the same code, as there), then one walk down from the first-level codes."""
    th_hat = np.zeros((len(l1p), len(model.labelmap)))
    for d, (first, second, third) in enumerate(zip(l1p, l2p, l3p)):
        load = {}
        for group in list(third) + list(second) + [first]:
            load.update(group)
        codes = list(load)
        for top, _ in first:
            for mid in _codes_below(top, codes):
                load[mid] *= load[top]
                for c in codes:                                # every code that holds  mid + one more digit
                    for i in range(len(c) - len(mid)):
                        if c.startswith(mid, i) and c[i + len(mid)].isdigit():
                            load[c[i:i + len(mid) + 1]] *= load[mid]
        th_hat[d, [model.labelmap[c] for c in codes]] = [load[c] for c in codes]
    return th_hat
