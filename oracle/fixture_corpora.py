"""Seeded toy corpora behind tests/golden/tiny_*.npz  --  TEST INFRASTRUCTURE.

Kept apart from gen_golden.py (which imports the reference and therefore only runs in the build
container) so that tests can rebuild the very same token/label lists and drive the drop-in classes
through their public constructors.
"""
import numpy as np


def synth_corpus(rng, D, V, n_labels, max_labs, len_lo, len_hi):
    vocab = ["w%04d" % i for i in range(V)]
    pz = 1.0 / np.arange(1, V + 1)
    pz /= pz.sum()
    docs, labs = [], []
    labelset = ["L%03d" % i for i in range(n_labels)]
    for d in range(D):
        n = int(rng.integers(len_lo, len_hi + 1))
        docs.append([vocab[i] for i in rng.choice(V, size=n, p=pz)])
        nl = int(rng.integers(0, max_labs + 1)) if n_labels else 0
        labs.append([labelset[i] for i in rng.choice(n_labels, size=min(nl, n_labels), replace=False)]
                    if nl else [])
    return docs, labs, labelset


TINY = [
    # name        D    V   labels max_labs  lens      alpha beta  sweeps
    ("k05",      40,  60,    4,   3,      (3, 25),   0.1, 0.01, 3),    # K<8: sequential np.sum
    ("k12",      60, 120,   11,   4,      (5, 40),   0.1, 0.01, 3),    # one row + tail
    ("k20dense", 50, 100,   19,  19,      (5, 40),   0.5, 0.1,  3),    # Cascade-root sized
    ("k40",      60, 150,   39,   6,      (5, 50),   0.1, 0.01, 3),
    ("k90",      40, 150,   89,  20,      (10, 50),  0.1, 0.01, 2),    # 12 slots per lane
    ("k128",     40, 200,  127,  30,      (10, 60),  0.1, 0.01, 2),    # exactly one full leaf
    ("k130",     40, 200,  129,  30,      (10, 60),  0.1, 0.01, 2),    # two leaves, tail 2
    ("k200",     30, 150,  199,  40,      (10, 60),  0.001, 0.001, 2),
    ("k392",     30, 150,  391,   7,      (10, 60),  0.1, 0.01, 2),    # abstracts-shaped, 4 leaves
    ("k512",     24, 150,  511, 200,      (10, 60),  0.1, 0.01, 2),    # 4 full leaves
    ("k777",     16, 120,  776, 300,      (10, 50),  0.1, 0.01, 2),    # unbalanced tree
    ("k1024",    12, 100, 1023, 400,      (10, 40),  0.1, 0.01, 2),    # the largest narrow K: 8 full leaves, 64 lanes x 16 slots
    # every label in every document (ALL_LABELS below): a dense mask with K == KP -- the dense 16-slot kernels with the commit log,
    # i.e. the 16-bit rows at four waves per SIMD the bench's headline runs on, against the reference itself
    ("k512dense",  30, 150,  511, 511,    (10, 60),  0.1, 0.01, 2),
    ("k1024dense", 12, 100, 1023, 1023,   (10, 40),  0.1, 0.01, 2),
    # ... and the narrower forms of the same kernel: K = 256 / 128 with eight / sixteen documents per wavefront (D is no multiple of
    # either: the last wavefront and the last workgroup are partly empty)
    ("k256dense",  45, 150,  255, 255,    (10, 60),  0.1, 0.01, 2),
    ("k128dense",  70, 150,  127, 127,    (10, 60),  0.1, 0.01, 2),
    # ... with positions that hold no topic (K < KP): one leaf of 100 (tail 4), two leaves 96 + 104, four leaves 96 + 104 + 96 + 104;
    # k250dense (three leaves 120 + 64 + 66 in a four-leaf layout, an unbalanced tree with a tail) stays on the general kernel
    ("k100dense",  70, 150,   99,  99,    (10, 60),  0.1, 0.01, 2),
    ("k200dense",  45, 150,  199, 199,    (10, 60),  0.1, 0.01, 2),
    ("k250dense",  30, 150,  249, 249,    (10, 60),  0.1, 0.01, 2),
    ("k400dense",  30, 150,  399, 399,    (10, 60),  0.1, 0.01, 2),
    # wide layouts (more than 8 pairwise leaves: 64-lane tiers of one wavefront)
    ("k1031",    10, 100, 1030, 400,      (10, 40),  0.1, 0.01, 2),    # 9 leaves -> 2 tiers x 16 slots, tail 7
    ("k1100",    10, 100, 1099,  30,      (10, 40),  0.1, 0.01, 2),    # 16 leaves, 12 slots per lane, tail 4, sparse labels
    ("k2100",     8,  80, 2099, 900,      (10, 40),  0.1, 0.01, 2),    # 22 leaves -> 3 tiers (not a power of two), tail 4
    ("k3000",     6,  80, 2999, 1500,     (10, 30),  0.5, 0.1,  2),    # 32 leaves -> 4 tiers x 12 slots, no tail
]


ALL_LABELS = ("k512dense", "k1024dense", "k256dense", "k128dense", "k100dense", "k200dense", "k250dense", "k400dense")      # fixtures whose documents carry the whole label set


def all_labels(name, docs, labs, labelset):
    """labs of fixture `name`: every label in every document for the ALL_LABELS fixtures, else as drawn"""
    return [list(labelset) for _ in docs] if name in ALL_LABELS else labs


def tiny_corpus(name):
    """-> (docs, labs, labelset, alpha, beta, sweeps, numpy_seed) of fixture tiny_<name>."""
    for (nm, D, V, nl, ml, (lo, hi), alpha, beta, sweeps) in TINY:
        if nm == name:
            rng = np.random.default_rng(sum(map(ord, name)))
            docs, labs, labelset = synth_corpus(rng, D, V, nl, ml, lo, hi)
            return docs, all_labels(name, docs, labs, labelset), labelset, alpha, beta, sweeps, 1000 + len(name)
    raise KeyError(name)


def cascade_corpus(seed=11, D=90, V=140):
    """Toy hierarchical corpus in the shape CascadeLDA.load_corpus produces: every document's label
    list holds ALL prefixes of its 3-character codes (e.g. 'B', 'B2', 'B21')."""
    rng = np.random.default_rng(seed)
    codes = ["A11", "A12", "A21", "A22", "A23", "B11", "B21", "B22", "C31", "C32", "C33"]
    vocab = ["t%04d" % i for i in range(V)]
    pz = 1.0 / np.arange(1, V + 1)
    pz /= pz.sum()
    docs, labs, seen = [], [], {}
    for _ in range(D):
        n = int(rng.integers(8, 45))
        docs.append([vocab[i] for i in rng.choice(V, size=n, p=pz)])
        picked = [codes[i] for i in rng.choice(len(codes), size=int(rng.integers(1, 4)), replace=False)]
        lab = []
        for c in picked:
            for p in (c[:1], c[:2], c[:3]):
                if p not in lab:
                    lab.append(p)
        for x in lab:
            seen.setdefault(x, 1)
        labs.append(lab)
    return docs, labs, list(seen.keys())
