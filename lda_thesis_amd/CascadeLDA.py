"""Drop-in module for the reference's ``CascadeLDA.py``: an ensemble of small Labeled-LDA problems,
one per parent node of the label tree, each trained by the HIP Gibbs sweep.

Surface mirrored (``from CascadeLDA import *`` in /root/reference/evaluate_CascadeLDA.py:1, which also
relies on ``np`` and ``re`` leaking through): ``load_corpus``, ``partition_label``, ``CascadeLDA``,
``SubLDA``, ``split_data``, ``prune_dict``, ``train_it``.

Scheduling (BASELINE.json configs[4]): the 1 + 19 + 102 sub-problems of the abstracts corpus share
nothing but the read-only corpus, so with ``torch.distributed`` initialised they are spread over the
ranks (longest-processing-time first, one sub-problem at a time per GPU) and the rows of ``ph`` are
combined at the end; there is no collective while training ("replicas only").
"""
import csv
import re
import sys

import numpy as np

from . import text as _text
from .corpus import csr_from_doc_tups
from .sampler import GibbsSampler, HostOrDevice

__all__ = ["np", "re", "load_corpus", "partition_label", "CascadeLDA", "SubLDA", "split_data",
           "prune_dict", "train_it"]

_JEL = re.compile(r"[A-Z]\d{2}")


def partition_label(lab, d):
    """'E32' -> ['E', 'E3', 'E32'] (all prefixes up to depth d): reference CascadeLDA.py:52-53."""
    return [lab[:i + 1] for i in range(d)]


def load_corpus(filename, d=3):
    """As LabeledLDA.load_corpus but every label is expanded to all its prefixes
    (reference CascadeLDA.py:8-49)."""
    limit = sys.maxsize
    while True:
        try:
            csv.field_size_limit(limit)
            break
        except OverflowError:
            limit //= 10
    texts, labs, seen = [], [], {}
    with open(filename, "r") as fh:
        for row in csv.reader(fh):
            field = row[2]
            if len(field) > 3:
                parts = [p for tok in field.split(" ") if _JEL.search(tok) for p in partition_label(tok, d)]
                lab = list(set(parts))
            else:
                lab = partition_label(field, d)
            for x in lab:
                seen.setdefault(x, 1)
            texts.append(row[1])
            labs.append(lab)
    print("Stemming documents ....")
    return _text.preprocess_documents(texts), labs, list(seen.keys())


class SubLDA(object):
    """One Labeled-LDA sub-problem (reference CascadeLDA.py:347-434).  ``docs`` are doc2bow tuples.

    The reference's count initialisation (CascadeLDA.py:382-385) indexes ``n_k_v`` with the whole
    (word id, frequency) tuple, so column ``frequency`` is bumped as well as column ``word id``.
    Those phantom counts feed the sampler and ``get_ph`` for the rest of training; they are reproduced
    here on the host and uploaded as the initial ``n_kw``."""

    def __init__(self, docs, labs, labelset, dicti, alpha=0.001, beta=0.001, seed=None, stream_id=0,
                 device=None, defer=False):
        labelset.insert(0, "root")
        self.labelmap = {lab: i for i, lab in enumerate(labelset)}
        self.K = len(self.labelmap)
        self.dicti = dicti
        self.lablist = labelset
        self.alpha = alpha
        self.beta = beta
        self.labs = np.array([self.set_label(lab) for lab in labs])
        self.doc_tups = docs
        self.V = len(dicti)
        self.D = len(docs)
        self._ph = HostOrDevice(np.zeros((self.K, self.V), dtype=float))

        self.docs, self.freqs, z0 = [], [], []
        for doc, lab in zip(self.doc_tups, self.labs):
            if not doc:
                raise ValueError("not enough values to unpack: a document has no in-vocabulary word")
            ids, freqs = zip(*doc)
            self.docs.append(list(ids))
            self.freqs.append(list(freqs))
            z0.append(np.random.choice(self.K, size=len(doc), p=lab / lab.sum()))
        self._z0 = np.concatenate(z0) if z0 else np.zeros(0, dtype=np.int64)
        self.seed = seed
        self.stream_id = stream_id
        self._device = device
        self._sampler = None
        self.n_sites = int(self._z0.shape[0])
        if not defer:
            self._upload()

    def _initial_counts(self):
        """host statement of the reference's count initialisation incl. the phantom columns (CascadeLDA.py:382-385);
        the device path of _upload builds the same counts with llda_count_init + add_word_topic_counts."""
        doc_off, word, freq = csr_from_doc_tups(self.doc_tups)
        z = self._z0
        f64 = freq.astype(np.int64)
        if f64.size and int(f64.max()) >= self.V:
            raise IndexError("index %d is out of bounds for axis 1 with size %d" % (int(f64.max()), self.V))
        rows = np.repeat(np.arange(self.D), np.diff(doc_off))
        n_zk = np.bincount(z, weights=f64, minlength=self.K).astype(np.int64)
        n_d_k = np.zeros((self.D, self.K), dtype=np.int64)
        np.add.at(n_d_k, (rows, z), f64)
        n_k_v = np.zeros((self.K, self.V), dtype=np.int64)
        np.add.at(n_k_v, (z, word), f64)
        ghost = freq != word                     # n_k_v[z, (id, f)] += f touches column f too (once if f == id)
        np.add.at(n_k_v, (z[ghost], freq[ghost]), f64[ghost])
        return (doc_off, word, freq), dict(n_d_k=n_d_k, n_k_v=n_k_v, n_zk=n_zk)

    def _upload(self):
        if self._sampler is None:
            if self.seed is None:
                self.seed = int(np.random.randint(0, 2 ** 31 - 1))
            doc_off, word, freq = csr_from_doc_tups(self.doc_tups)
            if freq.size and int(freq.max()) >= self.V:
                raise IndexError("index %d is out of bounds for axis 1 with size %d" % (int(freq.max()), self.V))
            # counts from the assignments on the device (llda_count_init), then the phantom counts: the
            # reference's n_k_v[z, (id, f)] += f also bumps column f (once if f == id)
            self._sampler = GibbsSampler(doc_off, word, freq, self._z0, self.K, self.V, self.alpha,
                                         self.beta, labs=self.labs, counts=None, seed=self.seed,
                                         stream_id=self.stream_id, device=self._device, sharded=False)
            ghost = freq != word
            self._sampler.add_word_topic_counts(freq[ghost], self._z0[ghost], freq[ghost])
        return self._sampler

    @property
    def n_zk(self):
        return self._upload().n_zk()

    @property
    def n_d_k(self):
        return self._upload().n_d_k()

    @property
    def n_k_v(self):
        return self._upload().n_k_v()

    @property
    def z_dn(self):
        return self._upload().z_dn()

    def set_label(self, label):
        vec = np.zeros(len(self.labelmap))
        vec[0] = 1.0
        for x in label:
            vec[self.labelmap[x]] = 1.0
        return vec

    # running mean of get_ph(): a numpy array when read (reference CascadeLDA.py:372), on the device between
    # the read-outs of run_training
    @property
    def ph(self):
        return self._ph.get()

    @ph.setter
    def ph(self, value):
        self._ph.set(value)

    def get_ph(self):
        """row-normalised n_k_v without smoothing (phantom columns included): CascadeLDA.py:394-395
        (llda_readout_phi with the integer row sums as denominators)."""
        return self._upload().ph_rows().cpu().numpy()

    def training_iteration(self):
        """One Gibbs sweep: reference CascadeLDA.py:397-421 (same body as LabeledLDA.py:101-125); the status word as in
        LabeledLDA.training_iteration."""
        sm = self._upload()
        sm.sweep()
        sm.post_status()

    def run_training(self, it=120, thinning=15):
        """reference CascadeLDA.py:423-434: snapshot ph when (i+1)/thinning is integral."""
        for i in range(it):
            self.training_iteration()
            s = (i + 1) / thinning
            if s == int(s):
                print("Training iteration #", i + 1)
                sm = self._sampler
                sm.check_status()
                first = not s > 1
                m = (s - 1) / s
                sm.ph_rows(self._ph.on_device(sm.device, (self.K, self.V), fresh=first),
                           None if first else m, None if first else 1 - m)

    def release(self):
        """drop the device state (the ensemble driver keeps only ph)."""
        self.ph                                   # a running mean still on the device moves to the host
        self._sampler = None


class CascadeLDA(object):
    """Hierarchy of SubLDA problems (reference CascadeLDA.py:56-184)."""

    def __init__(self, docs, labs, labelset, dicti, alpha=0.001, beta=0.001, seed=None, device=None,
                 group=None):
        labelset.insert(0, "root")
        self.labelmap = {lab: i for i, lab in enumerate(labelset)}
        self.dicti = dicti
        self.K = len(self.labelmap)
        self.lablist = labelset
        self.alpha = alpha
        self.beta = beta
        self.vocab = list(dicti.values())
        self.w_to_v = dicti.token2id
        self.v_to_w = dicti.id2token
        self.labs = np.array([self.set_label(lab) for lab in labs])
        self.doc_tups = [dicti.doc2bow(x) for x in docs]
        self.docs, self.freqs = [], []
        for doc in self.doc_tups:
            if not doc:
                raise ValueError("not enough values to unpack: a document has no in-vocabulary word")
            ids, freqs = zip(*doc)
            self.docs.append(ids)
            self.freqs.append(freqs)
        self.D = len(docs)
        self.V = len(self.vocab)
        self.ph = np.zeros((self.K, self.V), dtype=float)
        self.perplx = []
        self.l1 = [[x for x in lab if len(x) == 1] for lab in labs]
        self.l2 = [[x for x in lab if len(x) == 2] for lab in labs]
        self.l3 = [[x for x in lab if len(x) == 3] for lab in labs]
        self.lablist_l1 = [x for x in self.lablist if len(x) == 1]
        self.lablist_l2 = [x for x in self.lablist if len(x) == 2]
        self.lablist_l3 = [x for x in self.lablist if len(x) == 3]
        self.rawlabs = labs
        self.seed = seed
        self._device = device
        self._group = group
        self._ensemble = None          # go_down_tree(keep_state=True): the trained batched ensemble
        self._batch_debug_margin = 0   # test hook of llda_sweep_batch (include/llda_gibbs.h)
        self._keep_subs = None         # tests: set to [] to keep the SubLDA objects of the sequential path
        self._owned_rows = np.zeros(0, dtype=np.int64)

    def set_label(self, label):
        vec = np.zeros(len(self.labelmap))
        vec[0] = 1.0
        for x in label:
            vec[self.labelmap[x]] = 1.0
        return vec

    def term_to_id(self, term):
        if term not in self.w_to_v:
            voca_id = len(self.vocab)
            self.w_to_v[term] = voca_id
            self.vocab.append(term)
        else:
            voca_id = self.w_to_v[term]
        return voca_id

    def sub_corpus(self, parent):
        """documents carrying ``parent`` with only its child labels kept: CascadeLDA.py:113-127."""
        level = len(parent)
        lab_level = self.l2 if level == 1 else self.l3
        present = [i for i, lab in enumerate(self.rawlabs) if parent in lab]
        doc_tups = [self.doc_tups[p] for p in present]
        labs = [[x for x in lab_level[p] if x[:level] == parent] for p in present]
        labset = sorted(set(x for sub in labs for x in sub))
        return doc_tups, labs, labset

    def get_sub_ph(self, subdocs, sublabs, sublabset, it=150, thinning=12):
        sub = SubLDA(subdocs, sublabs, sublabset, self.dicti, alpha=self.alpha, beta=self.beta,
                     seed=self.seed, device=self._device)
        sub.run_training(it=it, thinning=thinning)
        return sub.get_ph()

    # ---- test time (reference CascadeLDA.py:186-344), on the device ----
    def _test_stream(self, labels):
        """RNG stream of one cascade_test call: depends on the level and on the first label."""
        from .foldin import CASCADE_STREAM
        return CASCADE_STREAM + self.K * len(labels[-1]) + self.labelmap[labels[0]]

    def prep4test(self, doc, ph, seed=None, stream_id=0, doc_id=None):
        """start state (ids, freqs, z_dn, n_dk) for loadings ``ph`` (CascadeLDA.py:186-208)."""
        from .foldin import cascade_fold_in, doc_key
        tup = self.dicti.doc2bow(doc)
        r = cascade_fold_in(ph, self.alpha, self.beta, [tup], 0, 1, self._seed(seed), stream_id,
                            [doc_key(tup) if doc_id is None else doc_id], device=self._device)
        ids, freqs = zip(*tup)
        return ids, freqs, list(r["z"][0]), r["n_dk"][0]

    def _seed(self, seed):
        if seed is not None:
            return seed
        if self.seed is None:
            self.seed = int(np.random.randint(0, 2 ** 31 - 1))
        return self.seed

    def cascade_test(self, doc, it, thinning, labels, seed=None, doc_id=None):
        """thinned document-topic average over the label subset ``labels`` (CascadeLDA.py:210-247)."""
        from .foldin import cascade_fold_in, doc_key
        ids = [self.labelmap[x] for x in labels]
        tup = self.dicti.doc2bow(doc)
        r = cascade_fold_in(self.ph[ids, :], self.alpha, self.beta, [tup], it, thinning, self._seed(seed),
                            self._test_stream(labels), [doc_key(tup) if doc_id is None else doc_id],
                            device=self._device)
        return r["th_hat"][0]

    @staticmethod
    def _head(th_hat, labels, threshold):
        """labels whose sorted loads are needed to pass ``threshold`` cumulative mass, with their loads."""
        loads = np.sort(th_hat)[::-1]
        n = sum(np.cumsum(loads) < threshold) + 1
        order = np.argsort(th_hat)[::-1][:n]
        return [labels[i] for i in order], loads[:n]

    @staticmethod
    def _head_many(th, labels, threshold):
        """``_head`` for every row of a (documents, labels) matrix with the sorts and cumulative sums done once for the
        whole matrix (numpy sorts / accumulates each row exactly as it does a 1-D array)."""
        th = np.asarray(th)
        if th.shape[0] == 0:
            return []
        loads = np.sort(th, axis=1)[:, ::-1]
        n = (np.cumsum(loads, axis=1) < threshold).sum(axis=1) + 1
        order = np.argsort(th, axis=1)[:, ::-1]
        return [([labels[i] for i in order[r, :n[r]]], loads[r, :n[r]]) for r in range(th.shape[0])]

    def test_down_tree(self, doc, it, thinning, threshold, seed=None, doc_id=None):
        """walk the label tree: level 1 over all one-character labels, then the children of every label
        kept at the level above (CascadeLDA.py:249-297).  Returns (level_1, level_2, level_3)."""
        kw = dict(seed=seed, doc_id=doc_id)
        labels = self.lablist_l1
        keep, loads = self._head(self.cascade_test(doc, it, thinning, labels, **kw), labels, threshold)
        level_1, level_2, level_3 = list(zip(keep, loads)), [], []
        if "root" in keep:
            keep.remove("root")
        for parent in keep:
            pat = re.compile("^" + parent + "[0-9]{1}$")
            labels = list(filter(pat.match, self.lablist))
            labels.insert(0, parent)
            keep2, loads2 = self._head(self.cascade_test(doc, it, thinning, labels, **kw), labels, threshold)
            level_2.append(list(zip(keep2, loads2)))
            if parent in keep2:
                keep2.remove(parent)
            for parent2 in keep2:
                pat = re.compile("^" + parent2 + "[0-9]{1}$")
                labels = list(filter(pat.match, self.lablist))
                labels.insert(0, parent2)
                keep3, loads3 = self._head(self.cascade_test(doc, it, thinning, labels, **kw), labels, threshold)
                level_3.append(list(zip(keep3, loads3)))
        return level_1, level_2, level_3

    def cascade_test_batch(self, docs, it, thinning, labels, seed=None, doc_ids=None, stream=None, defer=False,
                           bows=None, ph_dev=None, hold=False):
        """cascade_test for several documents against the SAME label subset: one llda_foldin launch, one lane
        group per document.  Row d equals cascade_test(docs[d], ...) -- the RNG is keyed by the document, not by
        its place in the batch.  defer=True: enqueue on ``stream`` and return the pending launch."""
        import torch
        from .foldin import cascade_fold_in, doc_key
        ids = [self.labelmap[x] for x in labels]
        tups = [self.dicti.doc2bow(doc) for doc in docs] if bows is None else bows     # (bows: doc2bow done by the caller)
        keys = [doc_key(t) for t in tups] if doc_ids is None else list(doc_ids)
        # ph_dev: self.ph on the device already (test_down_tree_batch uploads it once for all nodes of the tree)
        ph = self.ph[ids, :] if ph_dev is None else ph_dev.index_select(0, torch.from_numpy(np.asarray(ids, dtype=np.int64)).to(ph_dev.device))
        r = cascade_fold_in(ph, self.alpha, self.beta, tups, it, thinning, self._seed(seed),
                            self._test_stream(labels), keys, device=self._device, stream=stream, defer=defer, hold=hold)
        return r if (defer or hold) else r["th_hat"]

    def test_down_tree_batch(self, docs, it, thinning, threshold, seed=None, streams=32):
        """test_down_tree for a list of documents, level by level: all documents that reach the same node of the
        label tree are sampled in one launch (at most one launch per node instead of several per document), and
        the launches of one level -- each a latency-bound chain on a few compute units -- run concurrently on
        ``streams`` HIP streams.  Returns [test_down_tree(doc, ...) for doc in docs]."""
        import torch
        docs = list(docs)
        children = lambda parent: [parent] + list(filter(re.compile("^" + parent + "[0-9]{1}$").match, self.lablist))
        out = [[None, [], []] for _ in docs]
        pool = [torch.cuda.Stream() for _ in range(max(1, streams))] if torch.cuda.is_available() else [None]
        ph_dev = None
        if torch.cuda.is_available():        # the loadings go to the device ONCE; every node gathers its rows there
            dev = torch.device(self._device if self._device is not None else "cuda:%d" % torch.cuda.current_device())
            ph_dev = torch.from_numpy(np.ascontiguousarray(self.ph, dtype=np.float64)).to(dev)
        bows = [self.dicti.doc2bow(doc) for doc in docs]
        keys = None
        if bows:
            from .foldin import doc_key
            keys = [doc_key(b) if b else 0 for b in bows]

        def level(todo):
            """todo: parent -> documents.  Nodes whose label lists have the same length share ONE launch
            (foldin.cascade_fold_in_many); all uploads first, then the launches.  -> parent -> (labels, th rows)."""
            from .foldin import cascade_fold_in, cascade_fold_in_many
            groups = {}
            for parent, members in todo.items():
                labels = children(parent)
                groups.setdefault(len(labels), []).append((parent, labels, members))
            held = []
            for i, (K_sub, nodes) in enumerate(groups.items()):
                if ph_dev is None:                      # (no device: the one-launch-per-node path reports the error)
                    for parent, labels, members in nodes:
                        held.append(([(parent, labels)], None, self.cascade_test_batch(
                            None, it, thinning, labels, seed=seed, defer=True, hold=True,
                            bows=[bows[d] for d in members], doc_ids=[keys[d] for d in members])))
                    continue
                jobs = []
                for parent, labels, members in nodes:
                    ids = torch.from_numpy(np.asarray([self.labelmap[x] for x in labels], dtype=np.int64)).to(ph_dev.device)
                    jobs.append((ph_dev.index_select(0, ids), [bows[d] for d in members], self._test_stream(labels),
                                 [keys[d] for d in members]))
                held.append(([(p_, l_) for p_, l_, _ in nodes], cascade_fold_in_many(
                    jobs, self.alpha, self.beta, it, thinning, self._seed(seed), stream=pool[i % len(pool)]), None))
            out_level = {}
            waiting = [(nodes, many() if many is not None else None, single() if single is not None else None)
                       for nodes, many, single in held]
            for nodes, res_many, pend in waiting:
                if res_many is not None:
                    for (parent, labels), th_rows in zip(nodes, res_many()):
                        out_level[parent] = (labels, th_rows)
                else:
                    out_level[nodes[0][0]] = (nodes[0][1], pend.result()["th_hat"])
            return {parent: out_level[parent] for parent in todo}

        # level 1: every document against the one-character labels
        labels = self.lablist_l1
        th = self.cascade_test_batch(None, it, thinning, labels, seed=seed, bows=bows, doc_ids=keys, ph_dev=ph_dev)
        todo2 = {}                                      # parent -> documents that kept it, in visiting order
        for d, (keep, loads) in enumerate(self._head_many(th, labels, threshold)):
            out[d][0] = list(zip(keep, loads))
            if "root" in keep:
                keep.remove("root")
            for parent in keep:
                todo2.setdefault(parent, []).append(d)
        # level 2, then level 3: one launch per node that some document reached
        res2, todo3 = {}, {}
        for parent, (labels, th) in level(todo2).items():
            for (keep2, loads2), d in zip(self._head_many(th, labels, threshold), todo2[parent]):
                res2[(d, parent)] = list(zip(keep2, loads2))
                if parent in keep2:
                    keep2.remove(parent)
                for parent2 in keep2:
                    todo3.setdefault(parent2, []).append(d)
        res3 = {}
        for parent2, (labels, th) in level(todo3).items():
            for (keep3, loads3), d in zip(self._head_many(th, labels, threshold), todo3[parent2]):
                res3[(d, parent2)] = list(zip(keep3, loads3))
        # assemble in the order test_down_tree visits the nodes
        for d in range(len(docs)):
            for parent, _ in out[d][0]:
                if parent == "root":
                    continue
                lvl2 = res2[(d, parent)]
                out[d][1].append(lvl2)
                for parent2, _ in lvl2:
                    if parent2 != parent:
                        out[d][2].append(res3[(d, parent2)])
        return [tuple(x) for x in out]

    def run_test(self, docs, it, thinning, depth="all", seed=None):
        """flat test over all labels (or the labels of one depth): CascadeLDA.py:299-344."""
        from .foldin import CASCADE_STREAM, cascade_fold_in
        inds = None
        if depth in [1, 2, 3]:
            inds = np.where([len(x) in [depth, 4] for x in self.lablist])[0]
        elif depth == "all":
            inds = range(self.K)
        ph = self.ph[inds, :]
        if len(docs) and it < thinning:
            raise UnboundLocalError("local variable 'th' referenced before assignment")
        tups = [self.dicti.doc2bow(x) for x in docs]
        r = cascade_fold_in(ph, self.alpha, self.beta, tups, it, thinning, self._seed(seed),
                            CASCADE_STREAM + 0xFFFF, np.arange(len(tups)), flat=True, device=self._device)
        for _ in tups:
            for i in range(it):
                if (i + 1) % thinning == 0:
                    print("----")
                    print("Testing iteration #", i + 1)
        return r["th_hat"]

    # ---- ensemble driver ----
    def enumerate_subproblems(self):
        """All sub-problems in the reference's visiting order (CascadeLDA.py:135-184):
        root, then for every level-1 label l: l, then every level-2 label under l.
        Returns a list of dicts {parent, doc_tups, labs, labset}."""
        tasks = [dict(parent="root", doc_tups=self.doc_tups, labs=self.l1, labset=self.lablist_l1)]
        for l in [x for x in self.lablist_l1 if x != "root"]:
            doc_tups, labs, labset = self.sub_corpus(parent=l)
            tasks.append(dict(parent=l, doc_tups=doc_tups, labs=labs, labset=labset))
            for l2 in [x for x in self.lablist_l2 if x[0] == l]:
                doc_tups, labs, labset = self.sub_corpus(parent=l2)
                tasks.append(dict(parent=l2, doc_tups=doc_tups, labs=labs, labset=labset))
        return tasks

    def plan_subproblems(self):
        """Every sub-problem of go_down_tree in the reference's visiting order (CascadeLDA.py:135-184) as index
        arrays -- what enumerate_subproblems + SubLDA.__init__ build as python lists, without touching the corpus:
        dict(parent, labset (children, 'root' NOT yet inserted), K, docs (member document ids, ascending),
        allowed ((D_p, A_max) local topic ids of every member, ascending, -1 padded; topic 0 = the sub-problem's
        'root'), n_allowed)."""
        # self.labs[d, labelmap[x]] is 1 exactly when x is among the raw labels of document d (set_label), so `parent in lab`
        # (sub_corpus, CascadeLDA.py:115) and "the children of parent the document carries" are column reads of that matrix
        L = np.asarray(self.labs) != 0
        col = self.labelmap
        by_prefix = {}
        for x in self.lablist:
            if len(x) in (2, 3):
                by_prefix.setdefault(x[:-1], []).append(x)
        order = ["root"]
        for l in [x for x in self.lablist_l1 if x != "root"]:
            order.append(l)
            order += [x for x in self.lablist_l2 if x[0] == l]
        plans = []
        for parent in order:
            if parent == "root":
                docs = np.arange(self.D, dtype=np.int64)
                labset = [x for x in self.lablist_l1 if x != "root"]
                sub = L[:, [col[x] for x in labset]]
            else:
                docs = np.flatnonzero(L[:, col[parent]]).astype(np.int64)
                names = sorted(set(by_prefix.get(parent, [])))
                sub = L[docs][:, [col[x] for x in names]] if names else np.zeros((len(docs), 0), dtype=bool)
                present = sub.any(axis=0)
                labset = [x for x, here in zip(names, present) if here]      # sorted(set(...)) of the labels that occur
                sub = sub[:, present]
            K = 1 + len(labset)
            flags = np.ones((len(docs), K), dtype=bool)
            flags[:, 1:] = sub
            n_allowed = flags.sum(axis=1)
            a_max = int(n_allowed.max()) if len(docs) else 1
            # ascending local topic ids, -1 padded: argsort of ~flags is stable, allowed topics come first
            idx = np.argsort(~flags, axis=1, kind="stable")[:, :a_max]
            allowed = np.where(np.arange(a_max)[None, :] < n_allowed[:, None], idx, -1)
            plans.append(dict(parent=parent, labset=labset, K=K, docs=docs, allowed=allowed, n_allowed=n_allowed))
        return plans

    def go_down_tree(self, it, s, batched=True, keep_state=False):
        """Train every sub-problem and scatter its topic-word rows into ``self.ph``
        (reference CascadeLDA.py:135-184).  Sub-problem i (visiting order) uses RNG stream id i.

        batched=True: all sub-problems of this rank are trained TOGETHER (lda_thesis_amd/ensemble.py: a sweep of
        the whole ensemble is a handful of launches).  Priors below 1e-6, a sub-problem of more than 128 topics or a
        document that allows more than 64 of them -- outside what llda_sweep_batch covers -- take the one-by-one path
        instead, which gives the same result.  A rank that is assigned no sub-problem (more ranks than sub-problems)
        trains nothing and only takes part in the final all-reduce.
        keep_state=True keeps the trained ensemble in ``self._ensemble`` (tests)."""
        import torch.distributed as dist
        world, rank = 1, 0
        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(self._group), dist.get_rank(self._group)
        if self.seed is None:
            self.seed = int(np.random.randint(0, 2 ** 31 - 1))
        rng_state = np.random.get_state()
        done = False
        owner = None
        if batched and self.alpha >= 1e-6 and self.beta >= 1e-6 and self.V * self.beta < 2.0 ** 40:
            owner, done = self._go_down_tree_batched(it, s, world, rank, keep_state)     # done=False: a problem too wide
        if not done:
            np.random.set_state(rng_state)                  # the batched attempt consumed the same stream
            owner = self._go_down_tree_sequential(it, s, world, rank)
        if world > 1:
            import torch
            dev = self._device if self._device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
            # all-reduce a buffer that holds ONLY the rows this rank trained (rows are disjoint: sum = union);
            # rows left by an earlier call must not be summed world times
            mine = np.zeros_like(self.ph)
            rows = self._owned_rows
            mine[rows] = self.ph[rows]
            buf = torch.from_numpy(mine).to(dev)
            dist.all_reduce(buf, group=self._group)
            self.ph = buf.cpu().numpy()
        return owner

    def _label_rows(self, i, plan):
        """rows of self.ph the topics of sub-problem i fill (-1: not kept): the root problem keeps all its rows,
        every other one drops its local 'root' (CascadeLDA.py:143-144, 162-165, 181-184)."""
        rows = [self.labelmap[x] for x in plan["labset"]]
        return [self.labelmap["root"] if i == 0 else -1] + rows

    def _go_down_tree_batched(self, it, s, world, rank, keep_state):
        import torch
        from .ensemble import Ensemble
        plans = self.plan_subproblems()
        doc_off, word, freq = csr_from_doc_tups(self.doc_tups)
        lens = np.diff(doc_off)
        from .ensemble import MAX_BATCH_K, MAX_BATCH_ALLOWED, draw_initial_topics_device
        # decided from ALL plans, so that every rank takes the same path (the caller restores numpy's stream)
        if any(pl["K"] > MAX_BATCH_K or (len(pl["docs"]) and int(pl["n_allowed"].max()) > MAX_BATCH_ALLOWED) for pl in plans):
            return None, False
        # initial assignments of EVERY sub-problem, in visiting order, from numpy's global stream -- exactly the
        # uniforms the reference's per-document np.random.choice calls consume (CascadeLDA.py:373-381)
        n_docs = np.array([len(pl["docs"]) for pl in plans], dtype=np.int64)
        inst_len = lens[np.concatenate([pl["docs"] for pl in plans])]
        site_end = np.concatenate(([0], np.cumsum(inst_len)))[np.cumsum(n_docs)]
        sites = np.diff(np.concatenate(([0], site_end))).tolist()
        a_max = max(pl["allowed"].shape[1] for pl in plans)
        allowed_all = np.full((int(n_docs.sum()), a_max), -1, dtype=np.int64)
        row = 0
        for pl in plans:
            allowed_all[row:row + len(pl["docs"]), :pl["allowed"].shape[1]] = pl["allowed"]
            row += len(pl["docs"])
        n_allowed_all = np.concatenate([pl["n_allowed"] for pl in plans])
        u = np.random.random_sample(int(inst_len.sum()))            # (= the plans' draws one after the other)
        owner = lpt_assign(sites, world)
        mine = [i for i in range(len(plans)) if owner[i] == rank]
        if not mine:                                   # more ranks than sub-problems: nothing to train here
            self._owned_rows = np.zeros(0, dtype=np.int64)
            self._ensemble = None
            return owner, True

        def z_start(dev):                              # on the ensemble's device: the searchsorted of 1.3 M uniforms
            z_all = draw_initial_topics_device(allowed_all, n_allowed_all, inst_len, u, dev)
            if len(mine) < len(plans):                 # this rank's sub-problems only
                z_all = torch.cat([z_all[int(site_end[i]) - sites[i]:int(site_end[i])] for i in mine])
            return z_all
        ens = Ensemble([plans[i] for i in mine], z_start, doc_off, word, freq,
                       self.V, self.alpha, self.beta, self.seed, device=self._device, streams=mine)
        ens.debug_margin = self._batch_debug_margin
        for i in mine:
            if i > 0:
                print(" --- ")
                print("Working on parent node", plans[i]["parent"])
        for i in range(it):
            ens.sweep()
            if (i + 1) % s == 0:
                print("Training iteration #", i + 1)
        ens.check_status()
        ph_dev = torch.zeros((self.K, self.V), dtype=torch.float64, device=ens.device)
        label_rows = [self._label_rows(i, plans[i]) for i in mine]
        ens.scatter_ph(ph_dev, label_rows)
        rows = sorted(r for lr in label_rows for r in lr if r >= 0)
        self._owned_rows = np.asarray(rows, dtype=np.int64)
        self.ph[self._owned_rows] = ph_dev[torch.from_numpy(self._owned_rows).to(ens.device)].cpu().numpy()
        self._ensemble = ens if keep_state else None
        return owner, True

    def _go_down_tree_sequential(self, it, s, world, rank):
        tasks = self.enumerate_subproblems()
        # host initialisation of every sub-problem, in visiting order, on every rank (identical
        # consumption of numpy's global stream); device state only for the ones trained here
        subs = []
        for i, t in enumerate(tasks):
            subs.append(SubLDA(t["doc_tups"], t["labs"], t["labset"], self.dicti, alpha=self.alpha,
                               beta=self.beta, seed=self.seed, stream_id=i, device=self._device, defer=True))
        owner = lpt_assign([sub.n_sites for sub in subs], world)
        owned = []
        for i, (t, sub) in enumerate(zip(tasks, subs)):
            labset = t["labset"]                  # 'root' was inserted at position 0 by SubLDA
            if owner[i] == rank:
                if i > 0:
                    print(" --- ")
                    print("Working on parent node", t["parent"])
                sub.run_training(it=it, thinning=s)
                sub_ph = sub.get_ph()             # final state, as get_sub_ph returns (CascadeLDA.py:129-133)
                if self._keep_subs is not None:
                    self._keep_subs.append((i, sub))
                else:
                    sub.release()
                if i == 0:
                    ids = [self.labelmap[x] for x in labset]          # root problem keeps its 'root' row
                    self.ph[ids, :] = sub_ph
                else:
                    ids = [self.labelmap[x] for x in labset[1:]]
                    self.ph[ids, :] = sub_ph[1:, :]
                owned += ids
            labset.remove("root")
        self._owned_rows = np.asarray(sorted(owned), dtype=np.int64)
        return owner


def lpt_assign(costs, n_workers):
    """longest-processing-time-first assignment of tasks to workers; returns owner[task]."""
    load = [0] * n_workers
    owner = [0] * len(costs)
    for i in sorted(range(len(costs)), key=lambda j: (-costs[j], j)):
        w = min(range(n_workers), key=lambda k: (load[k], k))
        owner[i] = w
        load[w] += costs[i]
    return owner


def split_data(f="thesis_data.csv", d=3):
    a, b, c = load_corpus(f, d)
    zipped = list(zip(a, b))
    np.random.shuffle(zipped)
    a, b = zip(*zipped)
    split = int(len(a) * 0.9)
    return (a[:split], b[:split], c), (a[split:], b[split:], c)


def prune_dict(docs, lower=0.1, upper=0.9):
    dicti = _text.Dictionary(docs)
    dicti.filter_extremes(no_above=upper, no_below=int(lower * len(docs)))
    return dicti


def train_it(train_data, it=150, s=12, l=0.02, u=0.98, al=0.001, be=0.001):
    a, b, c = train_data
    dicti = prune_dict(a, lower=l, upper=u)
    cascade = CascadeLDA(a, b, c, dicti, alpha=al, beta=be)
    cascade.go_down_tree(it=it, s=s)
    return cascade
