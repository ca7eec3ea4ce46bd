"""Drop-in module for the reference's ``LabeledLDA.py`` with the Gibbs sweep on MI355X.

Surface mirrored (names reached through ``from LabeledLDA import *`` in
/root/reference/evaluate_LabeledLDA.py:1): ``load_corpus``, class ``LabeledLDA``, ``split_data``,
``prune_dict``, ``train_it``, ``test_it`` and the leaked ``np``.

What differs from the reference:
  * ``training_iteration()`` (reference LabeledLDA.py:101-125) is one launch of the HIP sweep kernel
    over all documents (per-document snapshot semantics, keyed Philox draw) instead of a python loop;
  * the sufficient statistics live in HBM; ``n_d_k``, ``n_k_v``, ``n_zk``, ``z_dn`` are properties that
    materialise them on the host in the reference's shapes and dtypes (LabeledLDA.py:73-79);
  * ``perplexity()`` (LabeledLDA.py:256-265) and the test-time fold-in sampler run on the device.
Text preparation uses ``lda_thesis_amd.text`` instead of gensim (not installable here).
"""
import csv
import re
import sys

import numpy as np

from . import text as _text
from .corpus import csr_from_doc_tups
from . import _native
from .sampler import GibbsSampler, HostOrDevice, shard_documents

__all__ = ["np", "load_corpus", "LabeledLDA", "split_data", "prune_dict", "train_it", "test_it"]

_JEL = re.compile(r"[A-Z]\d{2}")


def _world_size():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _gather_rows(local):
    """concatenate the per-rank slices of a document- or site-indexed array (read-out path, host side)."""
    if _world_size() == 1:
        return local
    import torch.distributed as dist
    parts = [None] * _world_size()
    dist.all_gather_object(parts, local)
    return np.concatenate(parts, axis=0)


def _raise_csv_limit():
    limit = sys.maxsize
    while True:
        try:
            csv.field_size_limit(limit)
            return
        except OverflowError:
            limit //= 10


def _parse_labels(field, d):
    """label column -> list of labels truncated to depth d (reference LabeledLDA.py:31-41)."""
    if len(field) > 3:
        return [tok[:d] for tok in field.split(" ") if _JEL.search(tok)]
    return [field[:d]]


def load_corpus(filename, d):
    """CSV rows (id, text, space separated JEL codes) -> (token lists, label lists, labelset).
    Same outputs as reference LabeledLDA.py:7-46; tokenisation by lda_thesis_amd.text."""
    _raise_csv_limit()
    texts, labs, seen = [], [], {}
    with open(filename, "r") as fh:
        for row in csv.reader(fh):
            lab = _parse_labels(row[2], d)
            for x in lab:
                seen.setdefault(x, 1)
            texts.append(row[1])
            labs.append(list(set(lab)))
    print("Stemming documents ....")
    return _text.preprocess_documents(texts), labs, list(seen.keys())


class LabeledLDA(object):
    """Labeled LDA trained by collapsed Gibbs sampling on the GPU.

    Constructor arguments as the reference (LabeledLDA.py:50); ``seed`` keys the device RNG (default:
    one draw from numpy's global stream after the initial assignments, so ``np.random.seed`` makes a
    whole run reproducible) and ``device`` picks the GPU."""

    def __init__(self, docs, labs, labelset, dicti, alpha, beta, seed=None, device=None):
        labelset.insert(0, "root")                    # the caller's list is extended, as in the reference
        self.labelmap = {lab: i for i, lab in enumerate(labelset)}
        self.K = len(self.labelmap)
        self.dicti = dicti
        self.alpha = alpha
        self.beta = beta
        self.vocab = list(dicti.values())
        self.w_to_v = dicti.token2id
        self.v_to_w = dicti.id2token
        # labs[d] = set_label(labs[d]) for every document at once (reference LabeledLDA.py:63,94-99: root + the document's labels;
        # an unknown label is a KeyError there too)
        self.D = len(docs)
        lab_len = np.fromiter(map(len, labs), dtype=np.int64, count=len(labs))
        lab_col = np.fromiter((self.labelmap[x] for lab in labs for x in lab), dtype=np.int64, count=int(lab_len.sum()))
        self.labs = np.zeros((len(labs), self.K))
        self.labs[:, 0] = 1.0
        self.labs[np.repeat(np.arange(len(labs)), lab_len), lab_col] = 1.0
        self.doc_tups = [dicti.doc2bow(x) for x in docs]
        self.V = len(self.vocab)
        self._ph_hat = HostOrDevice(np.zeros((self.K, self.V), dtype=float))
        self._th_hat = HostOrDevice(np.zeros((self.D, self.K), dtype=float))
        self.cur_perplx = []

        doc_off, word, freq = csr_from_doc_tups(self.doc_tups)
        self._doc_off = doc_off
        lens = np.diff(doc_off)
        if len(lens) and int(lens.min()) == 0:
            raise ValueError("not enough values to unpack: a document has no in-vocabulary word")
        word_l, freq_l, off_l = word.tolist(), freq.tolist(), doc_off.tolist()
        self.docs = [word_l[a:b] for a, b in zip(off_l[:-1], off_l[1:])]
        self.freqs = [freq_l[a:b] for a, b in zip(off_l[:-1], off_l[1:])]
        # Initial assignments.  The reference draws np.random.choice(K, size=len(doc), p=lab / lab.sum()) per document
        # (LabeledLDA.py:80-88); numpy's legacy choice is cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(random_sample(n), 'right'),
        # so ONE random_sample of all sites in document order consumes the global stream identically and the searchsorted becomes a
        # lookup in the cdf table of the document's label-set size (ensemble.draw_initial_topics, checked against np.random.choice
        # itself in tests/test_host_logic.py) -- the same z under np.random.seed, without 4 171 python-level calls.
        z0 = self._initial_topics(lens)
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))
        self.seed = seed
        # with torch.distributed initialised the documents are sharded over the ranks by site count
        # (every rank builds the same model object from the same data; it keeps only its slice on the GPU)
        self._bounds = shard_documents(doc_off, _world_size())
        lo, hi = self._bounds[_rank()], self._bounds[_rank() + 1]
        s0, s1 = int(doc_off[lo]), int(doc_off[hi])
        self._sampler = GibbsSampler(doc_off[lo:hi + 1] - doc_off[lo], word[s0:s1], freq[s0:s1], z0[s0:s1],
                                     self.K, self.V, alpha, beta, labs=self.labs[lo:hi], counts=None, seed=seed,
                                     doc_base=lo, device=device)

    # ---- state in the reference's shapes / dtypes ----
    # With torch.distributed initialised the documents are sharded over the ranks: n_d_k, z_dn, th_hat, get_theta()
    # and pickling (__getstate__) GATHER the per-rank slices and are therefore COLLECTIVE -- every rank must read
    # them (reading on one rank only, e.g. `if rank == 0: pickle.dump(model)`, blocks).  n_zk, n_k_v, ph_hat and
    # perplexity() are replicated / already reduced and can be read anywhere.
    # (materialising the state synchronises anyway: the status word is looked at first -- where the reference would have raised
    # inside the sweep, LabeledLDA.py:117-119, the caller gets the ValueError no later than here)
    @property
    def n_zk(self):
        self._sampler.check_status()
        return self._sampler.n_zk()

    @property
    def n_d_k(self):
        self._sampler.check_status()
        return _gather_rows(self._sampler.n_d_k())

    @property
    def n_k_v(self):
        self._sampler.check_status()
        return self._sampler.n_k_v()

    @property
    def z_dn(self):
        self._sampler.check_status()
        z = _gather_rows(self._sampler.z_topics())
        return [z[self._doc_off[d]:self._doc_off[d + 1]].copy() for d in range(self.D)]

    # running means of phi / theta: numpy arrays when read (reference LabeledLDA.py:65-66); between the
    # thinning read-outs of run_training they stay on the device
    @property
    def ph_hat(self):
        return self._ph_hat.get()

    @ph_hat.setter
    def ph_hat(self, value):
        self._ph_hat.set(value)

    @property
    def th_hat(self):
        return self._th_hat.get(_gather_rows)

    @th_hat.setter
    def th_hat(self, value):
        self._th_hat.set(value)

    def _initial_topics(self, lens):
        from .ensemble import draw_initial_topics
        if int(lens.sum()) == 0:
            return np.zeros(0, np.int64)
        n_allowed = self.labs.sum(axis=1).astype(np.int64)
        rows, cols = np.nonzero(self.labs)                       # row major: a document's allowed topics ascending
        first = np.zeros(self.D + 1, dtype=np.int64)
        np.cumsum(n_allowed, out=first[1:])
        allowed = np.full((self.D, int(n_allowed.max()) if self.D else 1), -1, dtype=np.int64)
        allowed[rows, np.arange(rows.shape[0]) - first[rows]] = cols
        u = np.random.random_sample(int(lens.sum()))
        return draw_initial_topics(allowed, n_allowed, np.repeat(np.arange(self.D), lens), u)

    def set_label(self, label):
        vec = np.zeros(len(self.labelmap))
        vec[0] = 1.0
        for x in label:
            vec[self.labelmap[x]] = 1.0
        return vec

    # ---- training ----
    def training_iteration(self):
        """One Gibbs sweep over every (document, word) site: reference LabeledLDA.py:101-125.  A site whose probabilities are all
        zero makes numpy raise at that site (LabeledLDA.py:117-119); here the kernels set a status bit, which a caller that loops this
        method sees a few sweeps later (an asynchronous copy, no synchronisation) and at the latest when it reads the counts."""
        self._sampler.sweep()
        self._sampler.post_status()

    def run_training(self, iters, thinning):
        """Sweep loop with thinning read-outs and running means: reference LabeledLDA.py:127-153.  phi, theta,
        their running means and the three guards are evaluated on the device (llda_readout_phi / _theta)."""
        import torch
        sm = self._sampler
        lo, hi = self._bounds[_rank()], self._bounds[_rank() + 1]
        for n in range(iters):
            self.training_iteration()
            print('Running iteration # %d ' % (n + 1))
            if (n + 1) % thinning != 0:
                continue
            sm.check_status()
            self.cur_perplx.append(self.perplexity())
            s = (n + 1) / thinning
            first = s == 1
            keep, share = (None, None) if first else ((s - 1) / s, 1 / s)
            flags = torch.zeros((1,), dtype=torch.int32, device=sm.device)
            sm.phi(self._ph_hat.on_device(sm.device, (self.K, self.V), fresh=first), keep, share, flags)
            sm.theta(self._th_hat.on_device(sm.device, (sm.D, self.K), fresh=first, rows=(lo, hi)), keep, share)
            bad = int(flags.item())
            if bad & _native.READOUT_NEGATIVE:
                raise ValueError('A negative value occurred in self.ph_hat while saving iteration %d ' % n)
            if bad & _native.READOUT_NAN:
                raise ValueError('A nan has creeped into ph_hat')
            if bad & _native.READOUT_NO_LOAD:
                raise ValueError('A word in dictionary has no z-value')

    # ---- read-outs ----
    def get_phi(self):
        """(n_k_v + beta) / (n_zk + V*beta): reference LabeledLDA.py:231-234 (llda_readout_phi)."""
        return self._sampler.phi().cpu().numpy()

    def get_theta(self):
        """(n_d_k + labs*alpha) / row sums: reference LabeledLDA.py:236-239 (llda_readout_theta)."""
        return _gather_rows(self._sampler.theta().cpu().numpy())

    def perplexity(self):
        """exp(-sum_sites log(phi[:, w] . theta_d) / #sites), sites unweighted by frequency
        (reference LabeledLDA.py:256-265); evaluated by the llda_loglik HIP kernel."""
        return self._sampler.perplexity()

    def topwords_per_topic(self, topwords=10):
        ph = self.get_phi()
        names = list(self.labelmap.keys())
        return [[names[k]] + [self.v_to_w[v] for v in np.argsort(-ph[k, :])[:topwords]]
                for k in range(self.K)]

    # ---- test time (reference LabeledLDA.py:155-212), on the device ----
    def prep4test(self, doc, seed=None, stream_id=None):
        """start state (ids, freqs, z_dn, n_dk) of one held-out token list: LabeledLDA.py:155-177."""
        from .foldin import TEST_STREAM, fold_in
        tups = self.dicti.doc2bow(doc)
        r = fold_in(self.ph_hat, self.alpha, [tups], 0, 1, self.seed if seed is None else seed,
                    TEST_STREAM if stream_id is None else stream_id)
        ids, freqs = zip(*tups)
        return ids, freqs, list(r["z"][0]), r["n_dk"][0]

    def run_test(self, newdocs, it, thinning, seed=None, stream_id=None):
        """thinned average of n_dk / sum(n_dk) over ``it`` fold-in sweeps per held-out document
        (LabeledLDA.py:179-212); all documents and sweeps in one llda_foldin launch."""
        from .foldin import TEST_STREAM, fold_in
        tups = [self.dicti.doc2bow(x) for x in newdocs]
        ph = self._ph_hat.dev if self._ph_hat.dev is not None else self.ph_hat     # still on the device after training
        r = fold_in(ph, self.alpha, tups, it, thinning, self.seed if seed is None else seed,
                    TEST_STREAM if stream_id is None else stream_id)
        return r["th_hat"]

    # ---- predictions ----
    def get_pred(self, single_th, n=5):
        names = np.array(list(self.labelmap.keys()))
        order = np.argsort(-single_th)[:n]
        loads = np.flip(np.sort(single_th), axis=0)[:n]
        return list(zip(names[order], loads))

    def get_preds(self, all_th, n=5):
        return [self.get_pred(all_th[d, :], n) for d in range(all_th.shape[0])]

    # ---- pickling: pull the device state to the host (evaluate_LabeledLDA.py:142-145 pickles the model)
    def __getstate__(self):
        self.ph_hat, self.th_hat                      # bring the running means to the host
        state = {k: v for k, v in self.__dict__.items() if k != "_sampler"}
        state["_host_state"] = dict(n_zk=self.n_zk, n_d_k=self.n_d_k, n_k_v=self.n_k_v,
                                    z=_gather_rows(self._sampler.z_topics()),
                                    sweeps_done=self._sampler.sweeps_done)
        return state

    def __setstate__(self, state):
        host = state.pop("_host_state")
        self.__dict__.update(state)
        doc_off, word, freq = csr_from_doc_tups(self.doc_tups)
        self._bounds = shard_documents(doc_off, _world_size())
        lo, hi = self._bounds[_rank()], self._bounds[_rank() + 1]
        s0, s1 = int(doc_off[lo]), int(doc_off[hi])
        counts = dict(n_d_k=host["n_d_k"][lo:hi], n_k_v=host["n_k_v"], n_zk=host["n_zk"])
        self._sampler = GibbsSampler(doc_off[lo:hi + 1] - doc_off[lo], word[s0:s1], freq[s0:s1], host["z"][s0:s1],
                                     self.K, self.V, self.alpha, self.beta, labs=self.labs[lo:hi], counts=counts,
                                     seed=self.seed, doc_base=lo)
        self._sampler.sweeps_done = host["sweeps_done"]


def split_data(f, d=2):
    """shuffle + 90/10 split: reference LabeledLDA.py:268-278 (the labelset list object is shared)."""
    a, b, c = load_corpus(f, d)
    zipped = list(zip(a, b))
    np.random.shuffle(zipped)
    a, b = zip(*zipped)
    split = int(len(a) * 0.9)
    return (a[:split], b[:split], c), (a[split:], b[split:], c)


def prune_dict(docs, lower=0.1, upper=0.9):
    dicti = _text.Dictionary(docs)
    dicti.filter_extremes(no_above=upper, no_below=lower * len(docs))
    return dicti


def train_it(traindata, it=30, s=3, al=0.001, be=0.001, l=0.05, u=0.95):
    a, b, c = traindata
    dicti = prune_dict(a, lower=l, upper=u)
    llda = LabeledLDA(a, b, c, dicti, al, be)
    llda.run_training(it, s)
    return llda


def test_it(model, testdata, it=500, thinning=25, n=5):
    known = set(model.vocab)
    testdocs = [[x for x in doc if x in known] for doc in testdata[0]]
    th_hat = model.run_test(testdocs, it, thinning)
    preds = model.get_preds(th_hat, n)
    th_hat = [[round(x, 4) for x in single_th] for single_th in th_hat]
    return th_hat, preds
