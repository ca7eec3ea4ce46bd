"""N > 1 host path on CPU: two gloo ranks, documents sharded by site count, per-sweep all-reduce of
the integer deltas, fold.  The device entry points are replaced by the C oracle (tests/helpers.py
OracleBackend) -- this checks the sharding / exchange / layout logic of GibbsSampler, not the kernel.
The state after every sweep must equal the single-process O3 golden of the reference.

The workers take ``hip=True`` from tests/test_gpu_multirank.py: the same checks with the REAL HIP kernels, every
rank on cuda:0, the gloo group reducing CUDA tensors."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port, hip):
    """process-group + backend of one worker: the C-oracle stand-in on CPU tensors, or (hip) the real library on
    cuda:0.  -> device string"""
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if hip == "nccl":                           # one GPU per rank, RCCL (only where the box has several GPUs)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        if hip:
            torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if hip:
        from lda_thesis_amd import _native
        import lda_thesis_amd.ensemble as E
        import lda_thesis_amd.sampler as S
        _native.lib()
        assert S._native is _native and E._native is _native      # nothing stands in for the HIP library
        return "cuda:%d" % (rank if hip == "nccl" else 0)
    import c_oracle
    from helpers import use_oracle_backend
    use_oracle_backend(c_oracle)
    return "cpu"


def _worker(rank, world, port, name, counts_mode, overlap, q, empty_last=False, hip=False):
    dev = _setup(rank, world, port, hip)
    from lda_thesis_amd.sampler import GibbsSampler, shard_documents
    g = load_golden(name)
    if name == "tiny_k12":
        GibbsSampler.PAIR_LIMIT = 20        # mix of int16-pair rows and int32 rows (hot words) in the exchange
    off = g["doc_off"]
    b = shard_documents(off, world)
    if empty_last:                              # the last rank holds no document at all
        b = shard_documents(off, world - 1) + [int(g["D"])]
    lo, hi = b[rank], b[rank + 1]
    s0, s1 = int(off[lo]), int(off[hi])
    counts = None
    if counts_mode == "given":          # global n_k_v / n_zk replicas, local n_d_k rows
        counts = dict(n_d_k=g["init_n_d_k"][lo:hi], n_k_v=g["init_n_k_v"], n_zk=g["init_n_zk"])
    s = GibbsSampler(off[lo:hi + 1] - off[lo], g["word"][s0:s1], g["freq"][s0:s1], g["init_z"][s0:s1],
                     int(g["K"]), int(g["V"]), float(g["alpha"]), float(g["beta"]), labs=g["labs"][lo:hi],
                     counts=counts, seed=int(g["seed"]), doc_base=lo, device=dev,
                     commit_log=counts_mode == "built",   # both commit paths
                     overlap_ranges=overlap)
    ok = (s.rows is not None) == (counts_mode == "built")       # every rank logs -> packed exchange rows
    if overlap > 1 and counts_mode == "built":                   # pipelined exchange: one set of rows per document range
        ok &= len(s._rows_list) == overlap
        if hi > lo:
            ok &= len(s._calls) >= 2 and len(s._item_bounds) == overlap + 1
    if name == "tiny_k12":
        ok &= 0 < int((s.row_off < 0).sum()) < s.V
    for i in range(int(g["sweeps"])):
        s.sweep()
        key = "o3_s%d_" % (i + 1)
        ok &= np.array_equal(s.n_k_v(), g[key + "n_k_v"])          # full replica on every rank
        ok &= np.array_equal(s.n_zk(), g[key + "n_zk"])
        ok &= np.array_equal(s.n_d_k(), g[key + "n_d_k"][lo:hi])
        ok &= np.array_equal(s.z_topics(), g[key + "z"][s0:s1])
    q.put((rank, bool(ok), hi - lo))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,counts_mode,overlap", [("tiny_k40", "built", 1), ("tiny_k392", "given", 1),
                                                      ("tiny_k12", "built", 1), ("tiny_k40", "built", 3),
                                                      ("tiny_k12", "built", 2), ("tiny_k392", "given", 2),
                                                      ("tiny_k1024", "built", 16), ("tiny_k1031", "built", 1),
                                                      ("tiny_k2100", "given", 2)])
def test_two_rank_sharded_sweeps_match_single_process_golden(name, counts_mode, overlap):
    """overlap > 1: the exchange is pipelined over document ranges (the rows of range i are all-reduced
    asynchronously while range i+1 is sampled) -- same state, bit for bit; ("given", 2): ranks that commit with
    atomics ignore the ranges; ("tiny_k1024", 16): more ranges than a rank has documents; k1031 / k2100: wide layouts
    (more than 8 pairwise leaves), both commit paths."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, counts_mode, overlap, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) == int(load_golden(name)["D"])


@pytest.mark.parametrize("overlap", [1, 2])
def test_rank_with_an_empty_shard_issues_the_same_collectives(overlap):
    """three ranks, the last one without documents: it logs nothing, yet it must build the exchange rows and issue
    one all-reduce per overlap range like the others (else the job hangs or sums the wrong buffers)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 3, port, "tiny_k40", "built", overlap, q, True)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert sorted(n for _, _, n in res)[0] == 0


def test_single_process_oracle_backend_matches_golden(c_oracle, monkeypatch):
    """sanity of the stand-in itself (world size 1, no process group)."""
    from helpers import OracleBackend, assert_state_equal
    import lda_thesis_amd.sampler as S
    from lda_thesis_amd.sampler import GibbsSampler
    monkeypatch.setattr(S, "_native", OracleBackend(c_oracle))
    g = load_golden("tiny_k130")
    s = GibbsSampler(g["doc_off"], g["word"], g["freq"], g["init_z"], int(g["K"]), int(g["V"]),
                     float(g["alpha"]), float(g["beta"]), labs=g["labs"], seed=int(g["seed"]), device="cpu",
                     commit_log=True)
    assert s.commit_log is not None and int(s.item_len.sum()) == s.S
    np.testing.assert_array_equal(s.n_k_v(), g["init_n_k_v"])
    for i in range(int(g["sweeps"])):
        s.sweep()
        assert_state_equal(g, "o3_s%d" % (i + 1), s.n_k_v(), s.n_d_k(), s.n_zk(), s.z_topics())


def _cascade_worker(rank, world, port, batched, q, hip=False, all_on_rank0=False):
    dev = _setup(rank, world, port, hip)
    from fixture_corpora import cascade_corpus
    import lda_thesis_amd.CascadeLDA as C
    from lda_thesis_amd.sampler import GibbsSampler
    from lda_thesis_amd.text import Dictionary

    class DevSampler(GibbsSampler):            # (the stand-in works on CPU tensors)
        def __init__(self, *a, **k):
            k.update(device=dev)
            super().__init__(*a, **k)
    C.GibbsSampler = DevSampler
    import lda_thesis_amd.ensemble as E

    class DevEnsemble(E.Ensemble):             # the batched ensemble, same
        def __init__(self, plans, z_local, *a, **k):
            k.update(device=dev)
            super().__init__(plans, z_local, *a, **k)
    E.Ensemble = DevEnsemble
    if all_on_rank0:                           # as if there were more ranks than sub-problems: rank 1 trains nothing
        C.lpt_assign = lambda costs, n_workers: [0] * len(costs)
    g = load_golden("cascade_toy")
    docs, labs, labelset = cascade_corpus()
    np.random.seed(int(g["np_seed"]))
    c = C.CascadeLDA(docs, labs, list(labelset), Dictionary(docs), float(g["alpha"]), float(g["beta"]), seed=int(g["seed"]))
    owner = c.go_down_tree(it=int(g["it"]), s=int(g["s"]), batched=batched, keep_state=True)
    ok = bool(np.array_equal(c.ph, g["ph"])) and (c._ensemble is not None) == (batched and not (all_on_rank0 and rank > 0))
    # a second call must not add the rows of the first call in again (the all-reduce sums only the owned rows)
    ph1 = c.ph.copy()
    np.random.seed(int(g["np_seed"]))
    c2 = C.CascadeLDA(docs, labs, list(labelset), Dictionary(docs), float(g["alpha"]), float(g["beta"]), seed=int(g["seed"]))
    c2.ph = ph1.copy()                          # stale rows from an earlier run
    c2.go_down_tree(it=int(g["it"]), s=int(g["s"]), batched=batched)
    ok &= bool(np.array_equal(c2.ph, g["ph"]))
    q.put((rank, ok, sorted(set(owner))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batched", [True, False])
def test_cascade_subproblems_spread_over_two_ranks(batched):
    """CascadeLDA.go_down_tree with torch.distributed: sub-problems LPT-assigned to 2 ranks (each rank trains its
    share as ONE batched ensemble, or one by one), disjoint ph rows unioned by one all-reduce; both ranks must end
    with the single-process reference ph."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cascade_worker, args=(r, 2, port, batched, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == [0, 1]                      # both ranks own some sub-problems


def _cascade_idle(rank, world, port, batched, q):
    _cascade_worker(rank, world, port, batched, q, all_on_rank0=True)


@pytest.mark.parametrize("batched", [True, False])
def test_cascade_rank_without_subproblems(batched):
    """a rank that owns no sub-problem builds no ensemble, takes part in the all-reduce and ends with the full ph."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cascade_idle, args=(r, 2, port, batched, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == [0]


def _llda_worker(rank, world, port, q, hip=False):
    dev = _setup(rank, world, port, hip)
    from fixture_corpora import tiny_corpus
    import lda_thesis_amd.LabeledLDA as L
    from lda_thesis_amd.sampler import GibbsSampler
    from lda_thesis_amd.text import Dictionary

    class DevSampler(GibbsSampler):
        def __init__(self, *a, **k):
            k.update(device=dev)
            super().__init__(*a, **k)
    L.GibbsSampler = DevSampler
    g = load_golden("tiny_k40")
    docs, labs, labelset, alpha, beta, sweeps, npseed = tiny_corpus("k40")
    np.random.seed(npseed)
    m = L.LabeledLDA(docs, labs, list(labelset), Dictionary(docs), alpha, beta, seed=int(g["seed"]))
    ok = m._sampler.D < m.D                           # this rank really holds a slice only
    for i in range(sweeps):
        m.training_iteration()
        key = "o3_s%d_" % (i + 1)
        ok &= np.array_equal(m.n_k_v, g[key + "n_k_v"]) and np.array_equal(m.n_d_k, g[key + "n_d_k"])
        ok &= np.array_equal(m.n_zk, g[key + "n_zk"]) and np.array_equal(np.concatenate(m.z_dn), g[key + "z"])
    ok &= bool(np.array_equal(m.get_phi(), g["o3_phi"]) and np.array_equal(m.get_theta(), g["o3_theta"]))
    # run_training: thinning read-outs and running means accumulate on each rank's slice, th_hat is gathered
    g = load_golden("runtraining_k12")
    docs, labs, labelset, alpha, beta, sweeps, npseed = tiny_corpus("k12")
    np.random.seed(npseed)
    m = L.LabeledLDA(docs, labs, list(labelset), Dictionary(docs), alpha, beta, seed=int(g["seed"]))
    m.run_training(int(g["iters"]), int(g["thinning"]))
    ok &= bool(np.array_equal(m.ph_hat, g["ph_hat"]) and np.array_equal(m.th_hat, g["th_hat"]))
    ok &= bool(np.allclose(np.array(m.cur_perplx), g["cur_perplx"], rtol=1e-9, atol=0))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_dropin_labeledlda_shards_documents_over_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_llda_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _flag_worker(rank, world, port, q):
    """a status flag raised on ONE rank (a site without a topic of positive probability, LabeledLDA.py:117-119) reaches check_status
    of EVERY rank after the next sweep's exchange: all ranks raise together instead of one raising and the others waiting in the
    next collective."""
    dev = _setup(rank, world, port, False)
    from lda_thesis_amd.sampler import GibbsSampler, shard_documents
    g = load_golden("tiny_k40")
    off = g["doc_off"]
    b = shard_documents(off, world)
    lo, hi = b[rank], b[rank + 1]
    s0, s1 = int(off[lo]), int(off[hi])
    res = []
    for commit in (True, False):                 # exchange rows / the int32 delta buffer
        s = GibbsSampler(off[lo:hi + 1] - off[lo], g["word"][s0:s1], g["freq"][s0:s1], g["init_z"][s0:s1], int(g["K"]), int(g["V"]),
                         float(g["alpha"]), float(g["beta"]), labs=g["labs"][lo:hi], seed=int(g["seed"]), doc_base=lo, device=dev,
                         commit_log=commit)
        s.sweep()
        s.check_status()                         # nothing flagged: nobody raises
        if rank == 1:
            s.status[0] |= 1                     # ... what the kernel sets
        s.sweep()
        try:
            s.check_status()
            res.append("no error")
        except ValueError:
            res.append("ValueError")
        ok = np.array_equal(s.n_k_v(), g["o3_s2_n_k_v"])        # the flag words did not leak into the counts
        res.append(bool(ok))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_status_flags_travel_with_the_deltas():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == res[1] == ["ValueError", True, "ValueError", True], res
