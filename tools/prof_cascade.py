

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
    for rep in range(3):
        np.random.seed(0)
        m = CascadeLDA(docs, labs, list(labelset), dicti, alpha=0.1, beta=0.01, seed=1)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            if rep == 2: pr.enable()
            m.go_down_tree(4, 2)
            if rep == 2: pr.disable()
        torch.cuda.synchronize()
        print("rep", rep, time.perf_counter() - t0)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
