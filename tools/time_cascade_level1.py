"""time the level-1 fold-in launch of CascadeLDA.test_down_tree_batch alone (464 held-out documents, 150 sweeps)."""
import io, os, sys, time
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
bows = [list(zip(tw[toff[d]:toff[d + 1]].tolist(), tf[toff[d]:toff[d + 1]].tolist())) for d in range(len(toff) - 1)]
bows = [b for b in bows if b]
print("docs", len(bows), "max sites", max(len(b) for b in bows), "mean", sum(len(b) for b in bows) / len(bows))
for it in (0, 150):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = m.cascade_test_batch(None, it, 25 if it else 1, m.lablist_l1, seed=1, bows=bows, doc_ids=list(range(len(bows))))
        torch.cuda.synchronize(); print("it", it, "level-1 launch + prep + readback: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
