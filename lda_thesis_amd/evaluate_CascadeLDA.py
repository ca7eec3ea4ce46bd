"""Command-line harness with the flags and the report of /root/reference/evaluate_CascadeLDA.py:144-224
(train the CascadeLDA ensemble on the GPU, walk every held-out document down the label tree, report
AUC / one-error / two-error / F1 per label depth).

    python -m lda_thesis_amd.evaluate_CascadeLDA -f abstracts_data.csv -i 4 -s 2
"""
import pickle
from optparse import OptionParser

import numpy as np

from .CascadeLDA import split_data, train_it
from .evaluate import binary_yreal, get_f1, macro_auc_roc, n_error, rates, setup_theta


def build_parser():
    p = OptionParser()
    p.add_option("-f", dest="file", help="dataset location")
    p.add_option("-d", dest="lvl", type="int", help="depth of label level", default=3)
    p.add_option("-i", dest="it", type="int", help="# of iterations - train and test")
    p.add_option("-s", dest="thinning", type="int", help="inter saving frequency", default=0)
    p.add_option("-a", dest="alpha", type="float", help="alpha prior", default=0.1)
    p.add_option("-b", dest="beta", type="float", help="beta prior", default=0.01)
    p.add_option("-l", dest="lower", type="float", help="lower threshold for dictionary pruning", default=0)
    p.add_option("-u", dest="upper", type="float", help="upper threshold for dictionary pruning", default=1)
    p.add_option("-p", action="store_true", dest="pickle", help="save pickle of model?", default=False)
    return p


def report(model, test, l1, l2, l3, depth, it, corpus_file):
    print("Model:               CascadeLDA")
    print("Corpus:             ", "Abstracts" if corpus_file == "thesis_data3.csv" else "Full texts")
    print("Label depth         ", depth)
    print("# of Gibbs samples: ", int(it))
    print("-----------------------------------")
    inds = np.where([len(x) == depth for x in model.labelmap.keys()])[0]
    y_bin = binary_yreal(test[1], model.labelmap)[:, inds]
    th_hat = setup_theta(l1, l2, l3, model)[:, inds]
    valid = np.intersect1d(np.where(th_hat.sum(axis=1) != 0)[0], np.where(y_bin.sum(axis=1) != 0)[0])
    y_bin, th_hat = y_bin[valid, :], th_hat[valid, :]
    tps, tns, fps, fns, fprs, tprs = rates(th_hat, y_bin)
    print("AUC ROC:                 ", macro_auc_roc(fprs, tprs))
    print("one error:               ", n_error(th_hat, y_bin, 1))
    print("two error:               ", n_error(th_hat, y_bin, 2))
    print("F1 score (macro average) ", get_f1(tps, fps, tns, fns))


def main(argv=None):
    opt, _ = build_parser().parse_args(argv)
    if opt.thinning == 0:
        opt.thinning = opt.it
    train, test = split_data(f=opt.file)            # the depth flag only limits the report, as in the reference
    model = train_it(train, it=opt.it, s=opt.thinning, l=opt.lower, u=opt.upper, al=opt.alpha, be=opt.beta)
    print("Testing test data, this may take a while")
    # (the reference calls test_down_tree document by document; the batch form returns the same lists with one
    # launch per node of the label tree)
    l1, l2, l3 = zip(*model.test_down_tree_batch(test[0], it=opt.it, thinning=opt.thinning, threshold=0.95))
    if opt.pickle:
        for name, obj in (("Cascade_model.pkl", model), ("Cascade_testset.pkl", test), ("Cascade_d1_pred.pkl", l1), ("Cascade_d2_pred.pkl", l2),
                          ("Cascade_d3_pred.pkl", l3)):
            pickle.dump(obj, open(name, "wb"))
        print("Saved the model and predictions as pickles!")
    for depth in range(1, int(opt.lvl) + 1):
        report(model, test, l1, l2, l3, depth, opt.it, opt.file)


if __name__ == "__main__":
    main()
