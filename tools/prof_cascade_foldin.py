"""cProfile of CascadeLDA.test_down_tree_batch on the abstracts fixture's held-out documents (where does the host time go)."""


def main():
    import cProfile, pstats, io, os, sys, time
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    from contextlib import redirect_stdout
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.corpus import cascade_corpus_from_csr
    from lda_thesis_amd.text import Dictionary
    g = np.load("tests/golden/abstracts_d3.npz")
    names = [str(x) for x in g["labelset"]]
    docs, labs, labelset = cascade_corpus_from_csr(g["doc_off"], g["word"], g["freq"], g["lab_off"], g["lab_idx"], names)
    dicti = Dictionary(docs)
    np.random.seed(0)
    m = CascadeLDA(docs, labs, list(labelset), dicti, alpha=0.1, beta=0.01, seed=1)
    with redirect_stdout(io.StringIO()):
        m.go_down_tree(4, 2)
    m.ph = np.nan_to_num(m.ph)
    toff, tw, tf = g["test_doc_off"], g["test_word"], g["test_freq"]
    held = []
    for d in range(len(toff) - 1):
        toks = []
        for v, f in zip(tw[toff[d]:toff[d + 1]], tf[toff[d]:toff[d + 1]]):
            toks += ["w%05d" % v] * int(f)
        if toks:
            held.append(toks)
    for rep in range(2):
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        if rep: pr.enable()
        trees = m.test_down_tree_batch(held, 150, 25, 0.95)
        if rep: pr.disable()
        print("rep", rep, len(held), "docs", time.perf_counter() - t0)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
    print(s.getvalue()[:5000])


if __name__ == "__main__":
    main()
