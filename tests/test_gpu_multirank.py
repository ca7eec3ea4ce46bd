"""N > 1 on hardware: several processes, each with its own HIP context on cuda:0, one gloo process group reducing
CUDA tensors -- the REAL ``llda_sweep`` + ``llda_commit_log`` -> exchange rows -> ``all_reduce`` -> ``llda_apply_rows``
(tests/test_distributed_gloo.py runs the same host logic with the C oracle standing in for the kernels; RCCL refuses
two ranks on one device, and the GPU box has one).  What the collective sums are the shared-count updates of the
reference (/root/reference/LabeledLDA.py:109-111,123-125).

(i)  goldens: the state of every rank after every sweep == the reference's O3 golden (2 and 4 ranks, one exchange
     per sweep and pipelined over 2 document ranges, narrow / wide layouts, both commit paths);
(ii) a 200 000-document slice of BASELINE configs[3] (K = 512, V = 100 000, 300 sites per document): every rank's
     [n_kw | n_k] replica, and the concatenation of the ranks' z / n_dk, equal the ONE-process run bit for bit;
(iii) the drop-in classes (LabeledLDA sharded over the ranks, CascadeLDA's sub-problems spread over them).
"""
import hashlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden
from test_distributed_gloo import _cascade_worker, _free_port, _llda_worker, _setup, _worker

pytestmark = pytest.mark.gpu


def _spawn(world, target, args, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + tuple(args) + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=timeout) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    return res


def _golden(rank, world, port, name, counts_mode, overlap, q):
    _worker(rank, world, port, name, counts_mode, overlap, q, hip=True)


@pytest.mark.parametrize("world,name,counts_mode,overlap", [
    (2, "tiny_k40", "built", 1), (2, "tiny_k40", "built", 2), (4, "tiny_k40", "built", 1), (4, "tiny_k40", "built", 2),
    (2, "tiny_k12", "built", 1), (4, "tiny_k12", "built", 2),            # int16-pair rows and int32 rows mixed
    (2, "tiny_k392", "given", 1), (4, "tiny_k392", "given", 2),         # atomics commit path: int32 delta buffer
    (2, "tiny_k512", "built", 2), (4, "tiny_k1024", "built", 1),        # the dense tiered kernel
    (2, "tiny_k1031", "built", 1), (2, "tiny_k2100", "built", 2),       # wide layouts
])
def test_hip_kernels_with_a_real_reduction_match_the_reference_golden(world, name, counts_mode, overlap):
    res = _spawn(world, _golden, (name, counts_mode, overlap))
    assert all(ok for _, ok, _ in res), res
    assert sum(n for _, _, n in res) == int(load_golden(name)["D"])


def _golden_empty(rank, world, port, overlap, q):
    _worker(rank, world, port, "tiny_k40", "built", overlap, q, empty_last=True, hip=True)


@pytest.mark.parametrize("overlap", [1, 2])
def test_hip_rank_without_documents(overlap):
    res = _spawn(3, _golden_empty, (overlap,))
    assert all(ok for _, ok, _ in res), res


def _golden_nccl(rank, world, port, name, counts_mode, overlap, q):
    _worker(rank, world, port, name, counts_mode, overlap, q, hip="nccl")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device "
                                                          "('Duplicate GPU detected')")
@pytest.mark.parametrize("name,counts_mode,overlap", [("tiny_k40", "built", 1), ("tiny_k12", "built", 2),
                                                      ("tiny_k392", "given", 1), ("tiny_k1031", "built", 1)])
def test_rccl_one_rank_per_gpu_matches_the_reference_golden(name, counts_mode, overlap):
    """the same check over RCCL with one GPU per rank -- runs only where the box has at least two GPUs"""
    world = min(4, torch.cuda.device_count())
    res = _spawn(world, _golden_nccl, (name, counts_mode, overlap))
    assert all(ok for _, ok, _ in res), res


# ------------------------------------------------------------------------------------------------ (ii) configs[3] slice
SLICE_DOCS, SLICE_BLOCK, SLICE_N, SLICE_V, SLICE_K, SLICE_SWEEPS = 200000, 12500, 300, 100000, 512, 3


def _digest(t):
    return hashlib.sha256(t.contiguous().cpu().numpy().tobytes()).hexdigest()


def _slice_worker(rank, world, port, overlap, rows16, q):
    _run_slice(rank, world, port, overlap, rows16, q, SLICE_DOCS, SLICE_N, SLICE_V, SLICE_K, SLICE_BLOCK)


K128 = dict(docs=40000, N=200, V=50000, K=128, block=5000)      # a slice of BASELINE configs[2]


def _k128_worker(rank, world, port, overlap, q):
    _run_slice(rank, world, port, overlap, None, q, K128["docs"], K128["N"], K128["V"], K128["K"], K128["block"])


def _run_slice(rank, world, port, overlap, rows16, q, SLICE_DOCS, SLICE_N, SLICE_V, SLICE_K, SLICE_BLOCK):
    from lda_thesis_amd.corpus import synthetic_corpus_blocks
    from lda_thesis_amd.sampler import GibbsSampler
    if world > 1:
        dev = _setup(rank, world, port, True)
    else:                                           # the single-process run: no process group at all
        _setup_paths_only()
        torch.cuda.set_device(0)
        dev = "cuda:0"
    lo, hi = SLICE_DOCS // world * rank, SLICE_DOCS // world * (rank + 1)
    doc_off, word, freq, z = synthetic_corpus_blocks(lo, hi, SLICE_N, SLICE_V, SLICE_K, 1234, dev, zipf_s=1.0,
                                                     block=SLICE_BLOCK)
    s = GibbsSampler(doc_off, word, freq, z, SLICE_K, SLICE_V, 0.1, 0.01, labs=None, seed=42, doc_base=lo, device=dev,
                     overlap_ranges=overlap, rows16=rows16)
    facts = dict(rows=s.rows is not None, logged=s.commit_log is not None, rows16=s.n_kw16 is not None, quad=bool(s.quad),
                 pair_rows=int((s.row_off[:-1] < 0).sum()) if s.row_off is not None else -1,
                 collectives=len(s._rows_list) if s._rows_list is not None else 0)
    for _ in range(SLICE_SWEEPS):
        s.sweep()
    s.check_status()
    torch.cuda.synchronize()
    nb = SLICE_BLOCK * SLICE_N
    blocks = [(lo // SLICE_BLOCK + b, _digest(s.z[b * nb:(b + 1) * nb]),
               _digest(s.n_dk[b * SLICE_BLOCK:(b + 1) * SLICE_BLOCK])) for b in range((hi - lo) // SLICE_BLOCK)]
    q.put((rank, _digest(s._counts), blocks, facts, int(s.n_k.sum(dtype=torch.int64))))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _setup_paths_only():
    import os
    import sys
    from conftest import ROOT
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


@pytest.fixture(scope="module")
def one_rank_slice():
    (rank, counts, blocks, facts, tokens), = _spawn(1, _slice_worker, (1, False))
    assert facts["logged"] and not facts["rows"]        # one process: the log is folded straight into n_kw
    assert tokens == SLICE_DOCS * SLICE_N
    return counts, blocks


# (rows16: every rank reads the rare words' rows from the 16-bit image of its n_kw replica, llda_sweep_args.n_kw16)
@pytest.mark.parametrize("world,overlap,rows16", [(2, 1, False), (2, 2, True), (4, 1, True), (4, 2, False)])
def test_configs3_slice_sharded_over_ranks_equals_the_one_rank_run(one_rank_slice, world, overlap, rows16):
    counts1, blocks1 = one_rank_slice
    res = _spawn(world, _slice_worker, (overlap, rows16), timeout=900)
    for rank, counts, blocks, facts, tokens in res:
        assert counts == counts1, "rank %d: [n_kw | n_k] differs from the one-rank run" % rank
        assert tokens == SLICE_DOCS * SLICE_N
        # the path under test: every rank logs, the exchange travels as packed rows (int16 pairs for most words,
        # int32 for the hot ones), one collective per document range
        assert facts["rows"] and facts["logged"] and 0 < facts["pair_rows"] < SLICE_V
        assert facts["collectives"] == overlap and facts["rows16"] == rows16
    got = sorted(b for _, _, blocks, _, _ in res for b in blocks)
    assert got == sorted(blocks1), "z / n_dk of the shards differ from the one-rank run"


def test_configs2_slice_on_the_sixteen_document_kernel_sharded_over_two_ranks():
    """K = 128 (a 40 000-document slice of BASELINE configs[2]): llda_sweep_quad_kernel<2> -- sixteen documents per wavefront, 16-byte
    site records, the 16-bit image of every rank's n_kw replica repacked each sweep -- under the exchange: both ranks' [n_kw | n_k] and
    the shards' z / n_dk equal the one-process run (which folds its log straight into n_kw)."""
    (rank, counts1, blocks1, facts1, tokens1), = _spawn(1, _k128_worker, (1,))
    assert facts1["quad"] and facts1["rows16"] and facts1["logged"] and tokens1 == K128["docs"] * K128["N"]
    res = _spawn(2, _k128_worker, (2,), timeout=900)
    for rank, counts, blocks, facts, tokens in res:
        assert counts == counts1, "rank %d: [n_kw | n_k] differs from the one-rank run" % rank
        assert facts["quad"] and facts["rows"] and facts["logged"] and facts["collectives"] == 2 and tokens == tokens1
    assert sorted(b for _, _, blocks, _, _ in res for b in blocks) == sorted(blocks1)


# ---- the same slice with SPARSE label sets (root + 7 labels per document): the sparse-label kernel, with and without its narrow image
SPARSE_DOCS = 100000


def _sparse_slice_worker(rank, world, port, image, q):
    from lda_thesis_amd.corpus import synthetic_corpus_blocks
    from lda_thesis_amd.sampler import GibbsSampler
    if world > 1:
        dev = _setup(rank, world, port, True)
    else:
        _setup_paths_only()
        torch.cuda.set_device(0)
        dev = "cuda:0"
    lo, hi = SPARSE_DOCS // world * rank, SPARSE_DOCS // world * (rank + 1)
    doc_off, word, freq, _ = synthetic_corpus_blocks(lo, hi, SLICE_N, SLICE_V, SLICE_K, 1234, dev, zipf_s=1.0, block=SLICE_BLOCK)
    labs, zs = [], []
    for b in range(lo // SLICE_BLOCK, hi // SLICE_BLOCK):            # label sets and start topics seeded per block of documents
        rng = np.random.default_rng(1000 + b)
        lab = np.sort(rng.integers(1, SLICE_K - 8, size=(SLICE_BLOCK, 7)), axis=1) + np.arange(7)
        lab = np.concatenate([np.zeros((SLICE_BLOCK, 1), dtype=lab.dtype), lab], axis=1)
        pick = rng.integers(0, 8, size=(SLICE_BLOCK, SLICE_N))
        labs.append(lab)
        zs.append(np.take_along_axis(lab, pick, axis=1).reshape(-1))
    lab, z = np.concatenate(labs), np.concatenate(zs)
    lab_off = np.arange(0, 8 * (hi - lo) + 1, 8, dtype=np.int64)
    s = GibbsSampler(doc_off, word, freq, z, SLICE_K, SLICE_V, 0.1, 0.01, labs=(lab_off, lab.reshape(-1)), seed=42, doc_base=lo,
                     device=dev, image=image)
    facts = dict(sparse=s.live_off is not None, image=0 if s.n_kw_img is None else 8 * s.n_kw_img.element_size(),
                 rows=s.rows is not None, logged=s.commit_log is not None)
    for _ in range(SLICE_SWEEPS):
        s.sweep()
    s.check_status()
    torch.cuda.synchronize()
    nb = SLICE_BLOCK * SLICE_N
    blocks = [(lo // SLICE_BLOCK + b, _digest(s.z[b * nb:(b + 1) * nb]),
               _digest(s.n_dk[b * SLICE_BLOCK:(b + 1) * SLICE_BLOCK])) for b in range((hi - lo) // SLICE_BLOCK)]
    q.put((rank, _digest(s._counts), blocks, facts, int(s.n_k.sum(dtype=torch.int64))))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def one_rank_sparse_slice():
    (rank, counts, blocks, facts, tokens), = _spawn(1, _sparse_slice_worker, (0,))
    assert facts["sparse"] and facts["image"] == 0 and tokens == SPARSE_DOCS * SLICE_N
    return counts, blocks


@pytest.mark.parametrize("world,image", [(1, 8), (1, 16), (1, None), (2, 8), (2, 16), (4, None)])
def test_sparse_label_slice_with_the_narrow_image_equals_the_plain_one_rank_run(one_rank_sparse_slice, world, image):
    """llda_sweep_args.n_kw_img on a 100 000-document slice of the sparse variant of configs[3]: every rank packs the image of ITS
    replica of n_kw after the exchange; states equal the one-process run without an image, bit for bit"""
    counts1, blocks1 = one_rank_sparse_slice
    res = _spawn(world, _sparse_slice_worker, (image,), timeout=900)
    for rank, counts, blocks, facts, tokens in res:
        assert counts == counts1, "rank %d: [n_kw | n_k] differs from the one-rank run without an image" % rank
        assert tokens == SPARSE_DOCS * SLICE_N and facts["sparse"] and facts["logged"]
        assert facts["image"] == (image if image is not None else facts["image"]) and facts["image"] in (8, 16)
        assert facts["rows"] == (world > 1)
    got = sorted(b for _, _, blocks, _, _ in res for b in blocks)
    assert got == sorted(blocks1), "z / n_dk of the shards differ from the one-rank run"


# ------------------------------------------------------------------------------------------------ (iii) drop-in classes
def _llda(rank, world, port, q):
    _llda_worker(rank, world, port, q, hip=True)


def test_hip_dropin_labeledlda_sharded_over_two_ranks():
    res = _spawn(2, _llda, ())
    assert all(ok for _, ok in res), res


def _cascade(rank, world, port, batched, q):
    _cascade_worker(rank, world, port, batched, q, hip=True)


@pytest.mark.parametrize("batched", [True, False])
def test_hip_cascade_subproblems_over_two_ranks(batched):
    res = _spawn(2, _cascade, (batched,))
    assert all(ok for _, ok, _ in res), res
    assert res[0][2] == [0, 1]
