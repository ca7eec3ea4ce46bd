"""Evaluation metrics of the reference's harness scripts, vectorised.

Mirrors /root/reference/evaluate_LabeledLDA.py:8-107 (one_roc, fpr_tpr, precision_recall, rates,
macro_auc_roc, n_error, get_f1, binary_yreal) and /root/reference/evaluate_CascadeLDA.py:95-127
(setup_theta).  Same names, same return shapes (python lists of per-threshold counts), same corner
cases (the counts are numpy integers as in the reference, so a rate with an empty denominator is
nan/inf with a RuntimeWarning rather than an exception, and get_f1 skips nan with nanmax).  Pinned by tests/golden/evaluate.npz.
"""
import re

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
    """mean over documents of the best F1 over thresholds: evaluate_LabeledLDA.py:85-93."""
    f1 = []
    for tp, fp, tn, fn in zip(tps, fps, tns, fns):
        prec, rec = precision_recall(tp, fp, tn, fn)
        with np.errstate(invalid='ignore'):
            raw = [(2 * p * r) / (p + r) for p, r in zip(prec, rec)]
        f1.append(np.nanmax(raw))
    return np.mean(f1)


def binary_yreal(label_strings, label_dict):
    y_true = np.zeros((len(label_strings), len(label_dict)), dtype=int)
    for d, lab in enumerate(label_strings):
        for l in lab:
            ind = label_dict.get(l)
            if ind is not None:
                y_true[d, ind] = 1
    return y_true


def setup_theta(l1p, l2p, l3p, model):
    """fold the per-level predictions of CascadeLDA.test_down_tree into one (docs, labels) matrix,
    multiplying every local load by the loads of its ancestors: evaluate_CascadeLDA.py:95-127."""
    n, k = len(l1p), len(model.labelmap)
    th_hat = np.zeros((n, k), dtype=float)
    for d in range(n):
        levels = dict()
        for tuplist in l3p[d]:
            levels.update(tuplist)
        for tuplist in l2p[d]:
            levels.update(tuplist)
        levels.update(l1p[d])
        lookup = " ".join(list(levels.keys()))
        for p in [s for (s, _) in l1p[d]]:
            for c in re.findall(re.compile("(" + p + "[0-9])(?:[^0-9]|$)"), lookup):
                levels[c] *= levels[p]
                for f in re.findall(re.compile(c + "[0-9]"), lookup):
                    levels[f] *= levels[c]
        labs, probs = zip(*levels.items())
        th_hat[d, [model.labelmap[x] for x in labs]] = probs
    return th_hat
