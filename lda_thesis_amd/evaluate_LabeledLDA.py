"""Command-line harness with the flags and the report of /root/reference/evaluate_LabeledLDA.py:110-180
(train Labeled LDA on the GPU, fold the held-out 10 % in, print AUC / one-error / two-error / F1).

    python -m lda_thesis_amd.evaluate_LabeledLDA -f abstracts_data.csv -d 3 -i 4 -s 4 -l 0 -u 1 -a 0.1 -b 0.01
"""
import pickle
from optparse import OptionParser

import numpy as np

from .evaluate import binary_yreal, get_f1, macro_auc_roc, n_error, rates
from .LabeledLDA import split_data, test_it, train_it


def build_parser():
    p = OptionParser()
    p.add_option("-f", dest="file", help="dataset location")
    p.add_option("-d", dest="lvl", type="int", default=3, help="depth of lab level")
    p.add_option("-i", dest="it", type="int", help="# of iterations")
    p.add_option("-s", dest="thinning", type="int", default=0, help="save frequency")
    p.add_option("-l", dest="lower", type="float", default=0, help="lower threshold for dictionary pruning")
    p.add_option("-u", dest="upper", type="float", default=1, help="upper threshold for dictionary pruning")
    p.add_option("-a", dest="alpha", type="float", default=0.1, help="alpha prior")
    p.add_option("-b", dest="beta", type="float", default=0.01, help="beta prior")
    p.add_option("-p", action="store_true", dest="pickle", default=False, help="Save the model as pickle?")
    return p


def report(model, test, th, lvl, it, corpus_file):
    print("Model:               Labeled LDA")
    print("Corpus:             ", "Abstracts" if corpus_file == "thesis_data3.csv" else "Full Texts")
    print("Label depth         ", lvl)
    print("# of Gibbs samples: ", int(it))
    print("-----------------------------------")
    y_bin = binary_yreal(test[1], model.labelmap)[:, 1:]        # the root label is in no label set
    th = th[:, 1:]
    keep = np.where(th.sum(axis=1) != 0)[0]                     # documents not assigned to 'root' entirely
    y_bin, th = y_bin[keep, :], th[keep, :]
    tps, tns, fps, fns, fprs, tprs = rates(th, y_bin)
    print("AUC ROC:                 ", macro_auc_roc(fprs, tprs))
    print("one error:               ", n_error(th, y_bin, 1))
    print("two error:               ", n_error(th, y_bin, 2))
    print("F1 score (macro average) ", get_f1(tps, fps, tns, fns))


def main(argv=None):
    opt, _ = build_parser().parse_args(argv)
    if opt.thinning == 0:
        opt.thinning = opt.it
    train, test = split_data(f=opt.file, d=opt.lvl)
    print("Starting training...")
    model = train_it(train, it=opt.it, s=opt.thinning, al=opt.alpha, be=opt.beta, l=opt.lower, u=opt.upper)
    print("Testing test data, this may take a while...")
    th, _ = test_it(model, test, it=opt.it, thinning=opt.thinning)
    th = np.array(th)
    if opt.pickle:
        pickle.dump(model, open("LabeledLDA_model.pkl", "wb"))
        pickle.dump(test, open("LabeledLDA_testset.pkl", "wb"))
        pickle.dump(th, open("LabeledLDA_theta.pkl", "wb"))
    report(model, test, th, opt.lvl, opt.it, opt.file)


if __name__ == "__main__":
    main()
