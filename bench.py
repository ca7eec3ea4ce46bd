"""bench.py -- Gibbs-sweep throughput of the HIP sampler on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload synth2|synth1|synth2_hostile|synth2_sparse|abstracts]

A "step" is one full Gibbs sweep: every site of every local document resampled once, plus the per-sweep
exchange (RCCL all-reduce of the n_kw / n_k deltas when N > 1) and the fold of the deltas into the counts.

Default workload = BASELINE.json configs[3], the configuration the metric is quoted on: ONE synthetic corpus of
1 000 000 documents x 300 sites, K = 512 dense label mask, V = 100 000 (Zipf word frequencies).  It fits one GPU
(16 GB of state), so N = 1 samples the whole corpus and N GPUs hold 1 000 000 / N documents each: STRONG scaling,
the corpus (generated in 64 seeded blocks) and therefore the state after every sweep are identical for every N.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>).

At N = 1 the same JSON line also carries (key "extra"):
  * "synth1"          BASELINE configs[2] (100k docs x 200 sites, K = 128): the roofline case BASELINE names;
  * "dense_k256"      the same corpus shape at K = 256 (the eight-documents-per-wavefront form of the 16-bit-row kernel; K = 128
                      runs sixteen, K = 512 four);
  * "hbm_bound"       a cache-hostile variant (uniform words, V = 500 000: n_kw = 1 GB > the 256 MB Infinity Cache)
                      that shows the genuinely HBM-bound regime of the same kernel;
  * "abstracts"       Labeled LDA on the tokenised abstracts_data.csv fixture (configs[0]/[1]) with its own
                      cpu_baseline -- the >= 50x target of BASELINE.json's north_star;
  * "sparse_labels", "cascade", "wide_k2048", "wide_sparse_k2048"   the sparse-label kernel on a big corpus, CascadeLDA's
                      ensemble (configs[4]) and the paths for K > 1024 (dense mask: one wavefront per document; sparse label
                      sets: the sparse-label kernel with the wide exact tier).
and the roofline of the dominant kernel from HBM-side PMC counters collected IN THIS RUN: the script re-runs itself
for a few sweeps under `rocprofv3 --kernel-trace --pmc ...` (separate passes for FETCH_SIZE, WRITE_SIZE and the SQ
group; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, the gfx950 correction of MI355X_MICROARCH.md).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from lda_thesis_amd.corpus import synthetic_corpus_blocks    # noqa: E402
from lda_thesis_amd.sampler import GibbsSampler              # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming copy)
N_SIMD = 1024                # 256 CUs x 4 SIMDs
N_XCD = 8                    # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (1.16e9 per 63 ms launch = 8 x 2.3 GHz)
MAX_CLOCK_HZ = 2.4e9         # MI355X_MICROARCH.md: max clock
SALU_CYCLES_PER_INST = 2.5   # fitted: profiles/r03_issue_model.md
VALU_CYCLES_PER_INST = 4     # SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.01 quad-cycles on this kernel (profiles/)

WORKLOADS = {
    # name: (documents in the corpus, sites per doc, V, K, zipf exponent, generator block, description)
    "synth2": (1000000, 300, 100000, 512, 1.0, 15625,
               "synthetic 1M docs x 300 tokens, K=512 dense mask, V=100k, doc-sharded over the GPUs of the job "
               "(BASELINE configs[3])"),
    "synth1": (100000, 200, 50000, 128, 1.0, 12500,
               "synthetic 100k docs x 200 tokens, K=128 dense mask, V=50k (BASELINE configs[2])"),
    "synth_k256": (100000, 200, 50000, 256, 1.0, 12500,
                   "synthetic 100k docs x 200 tokens, K=256 dense mask, V=50k (tools and tests: the eight-documents-per-wavefront form "
                   "of the 16-bit-row kernel)"),
    "synth_k100": (100000, 200, 50000, 100, 1.0, 12500,
                   "synthetic 100k docs x 200 tokens, K=100 dense mask, V=50k (tools: a layout with positions that hold no topic, KP = 128)"),
    "synth_k400": (100000, 300, 100000, 400, 1.0, 12500,
                   "synthetic 100k docs x 300 tokens, K=400 dense mask, V=100k (tools: four unequal leaves, KP = 512)"),
    "synth2_hostile": (125000, 300, 500000, 512, 0.0, 15625,
                       "cache-hostile variant of configs[3]: 125k docs x 300 tokens, K=512 dense mask, UNIFORM words over "
                       "V=500k -- n_kw is 1.02 GB, four times the Infinity Cache, every site reads a cold 2 KB row"),
    "synth2_sparse": (125000, 300, 100000, 512, 1.0, 15625,
                      "synthetic 125k docs x 300 tokens, K=512, sparse label mask (root + 7 random labels per doc), "
                      "V=100k (secondary variant of BASELINE configs[3])"),
    "synth2_sparse_hier": (125000, 300, 100000, 512, 1.0, 15625,
                           "synthetic 125k docs x 300 tokens, K=512, sparse label mask with CO-LOCATED labels: root + 7 labels drawn "
                           "from ONE block of 32 consecutive topic ids per document (a label hierarchy whose siblings have "
                           "neighbouring ids: 32 consecutive topics share a 128-byte line of every n_kw row), V=100k"),
    "synth2_sparse_scr": (125000, 300, 100000, 512, 1.0, 15625,
                          "the co-located label sets of synth2_sparse_hier with the topic ids SCRAMBLED by a fixed permutation: the same "
                          "co-occurrence structure, but the caller's label order no longer puts siblings next to each other -- what the "
                          "sampler's own column order of the narrow image (GibbsSampler(image_order=...)) is for"),
    "synth_wide_sparse": (125000, 300, 100000, 2048, 1.0, 15625,
                          "synthetic 125k docs x 300 tokens, K=2048 (a 'wide' layout), sparse label mask (root + 7 random "
                          "labels per doc), V=100k: Labeled LDA proper with thousands of labels -- the sparse-label kernel "
                          "with the wide exact tier"),
    "synth_wide": (20000, 100, 20000, 2048, 1.0, 2500,
                   "synthetic 20k docs x 100 tokens, K=2048 dense mask, V=20k: a 'wide' layout (16 pairwise leaves -> one "
                   "wavefront per document, DESIGN 4.7) -- the general path for K beyond the tuned kernels' 1024"),
    # real corpus: tokenised abstracts_data.csv, depth 3 (tests/golden/abstracts_d3.npz); sizes read from the file
    "abstracts": (4171, 0, 0, 392, 0.0, 0,
                  "Labeled LDA on abstracts_data.csv, depth 3, K=392 sparse label masks (BASELINE configs[0]/[1]); "
                  "replicated per GPU"),
}
ALPHA, BETA = 0.1, 0.01


def algorithmic_bytes(sites, docs, A):
    """per site 4*A + 32, per document 12*A + 16 (A = live topics)."""
    return sites * (4 * A + 32) + docs * (12 * A + 16)


# ------------------------------------------------------------------------------------------------ workloads
class one_device_lock(object):
    """cross-process mutex (flock on a file named after the rendezvous port) for the ranks of a --one-device run; re-usable"""

    def __init__(self):
        import tempfile
        self.path = os.path.join(tempfile.gettempdir(), "llda_bench_%s.lock" % os.environ.get("MASTER_PORT", "0"))
        self.fh = None

    def __enter__(self):
        import fcntl
        self.fh = open(self.path, "w")
        fcntl.flock(self.fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        torch.cuda.synchronize()                     # the section's GPU work is done before the next rank starts its own
        fcntl.flock(self.fh, fcntl.LOCK_UN)
        self.fh.close()
        self.fh = None
        return False


def build_sampler(name, dev, rank, world, dist_on, docs_total=0, docs_per_group=0, force_exchange=False,
                  overlap=None, rows16=None, build_lock=None):
    """-> (sampler, info dict).  Inputs are generated on the device."""
    Dt, N, V, K, zs, block, desc = WORKLOADS[name]
    if docs_total:
        Dt = docs_total
    info = dict(desc=desc, K=K, V=V, N=N, live_topics=float(K), docs_total=Dt)
    kw = {} if overlap is None else dict(overlap_ranges=overlap)
    if build_lock is not None:
        kw["build_lock"] = build_lock
    import contextlib
    locked = build_lock if build_lock is not None else contextlib.nullcontext()
    if rows16 is None and os.environ.get("LLDA_BENCH_ROWS16"):        # ablation: "on" / "off" for every sampler of the run
        rows16 = os.environ["LLDA_BENCH_ROWS16"] == "on"
    if rows16 is not None:
        kw["rows16"] = rows16
    if name == "abstracts":
        g = np.load(os.path.join(ROOT, "tests", "golden", "abstracts_d3.npz"))
        Dg, V, K = int(g["D"]), int(g["V"]), int(g["K"])
        # every rank samples its own replica of the corpus (independent chains: sharded=False)
        s = GibbsSampler(g["doc_off"], g["word"].astype(np.int32), g["freq"].astype(np.int32),
                         g["z_init"].astype(np.int64), K, V, ALPHA, BETA,
                         labs=(g["lab_off"], g["lab_idx"].astype(np.int64)), counts=None, seed=42 + rank,
                         device=dev, docs_per_group=docs_per_group, sharded=False)
        info.update(K=K, V=V, N=int(g["doc_off"][-1]) // Dg, live_topics=float(len(g["lab_idx"])) / Dg,
                    docs_total=Dg, docs_local=Dg, fixture=g)
        return s, info
    if Dt % (block * world):
        block = Dt // world if Dt % world == 0 else None
        if block is None:
            raise SystemExit("%d documents do not split over %d GPUs" % (Dt, world))
    lo, hi = Dt // world * rank, Dt // world * (rank + 1)
    with locked:
        doc_off, word, freq, z = synthetic_corpus_blocks(lo, hi, N, V, K, 1234, dev, zipf_s=zs, block=block)
    Dg = hi - lo
    info["docs_local"] = Dg
    if name in ("synth2_sparse", "synth_wide_sparse", "synth2_sparse_hier", "synth2_sparse_scr"):
        gen = torch.Generator(device=dev)
        gen.manual_seed(99 + rank)
        if name in ("synth2_sparse_hier", "synth2_sparse_scr"):
            # 7 distinct labels inside one block of 32 consecutive topic ids (block 0 without the root's id 0)
            blk = torch.randint(0, K // 32, (Dg, 1), device=dev, generator=gen)
            inner = torch.argsort(torch.rand((Dg, 31), device=dev, generator=gen), dim=1)[:, :7] + 1     # 7 of 1..31
            lab = torch.sort(blk * 32 + inner, dim=1).values
            if name == "synth2_sparse_scr":                  # the same sets under a fixed permutation of the ids 1 .. K-1
                g2 = torch.Generator(device=dev)
                g2.manual_seed(4242)
                perm = torch.cat([torch.zeros((1,), dtype=torch.int64, device=dev), torch.randperm(K - 1, device=dev, generator=g2) + 1])
                lab = torch.sort(perm[lab], dim=1).values
        else:
            # root + 7 distinct random labels per document: sort 7 draws, bump duplicates (still <= K-1)
            lab = torch.sort(torch.randint(1, K - 8, (Dg, 7), device=dev, generator=gen), dim=1).values
            lab = lab + torch.arange(7, device=dev)          # strictly increasing => distinct
        lab = torch.cat([torch.zeros((Dg, 1), dtype=lab.dtype, device=dev), lab], dim=1)
        pick = torch.randint(0, 8, (Dg * N,), device=dev, generator=gen)
        z = lab.repeat_interleave(N, dim=0)[torch.arange(Dg * N, device=dev), pick]
        lab_off = np.arange(0, 8 * Dg + 1, 8, dtype=np.int64)
        s = GibbsSampler(doc_off, word, freq, z, K, V, ALPHA, BETA, labs=(lab_off, lab.reshape(-1).cpu().numpy()),
                         counts=None, seed=42, doc_base=lo, device=dev, docs_per_group=docs_per_group, **kw)
        info.update(live_topics=8.0, lab=lab)
    else:
        s = GibbsSampler(doc_off, word, freq, z, K, V, ALPHA, BETA, labs=None, counts=None, seed=42,
                         doc_base=lo, device=dev, docs_per_group=docs_per_group,
                         exchange_always=bool(force_exchange), **kw)
    info.update(doc_off=doc_off, word=word, freq=freq)
    return s, info


CHECKSUM_FILE = os.path.join(ROOT, "profiles", "state_checksums.json")


def state_checksums(sampler):
    """digests of the shared counts in TOPIC order (independent of the device permutation): weighted sums of n_k and
    of the whole n_kw (int64, wrapping).  The state after s sweeps is the same for every number of GPUs, so an
    N > 1 run can be checked against the N = 1 values stored in profiles/state_checksums.json."""
    dev, K, V = sampler.device, sampler.K, sampler.V
    nk = sampler.n_k[sampler._topic_pos].to(torch.int64)
    c_k = int((nk * torch.arange(1, K + 1, device=dev)).sum().item())
    c_kw = 0
    step = max(1, (1 << 24) // K)
    wk = torch.arange(K, device=dev, dtype=torch.int64)
    for v0 in range(0, V, step):
        blk = sampler.n_kw[v0:v0 + step][:, sampler._topic_pos].to(torch.int64)
        idx = (torch.arange(v0, v0 + blk.shape[0], device=dev, dtype=torch.int64)[:, None] * K + wk[None, :]) % 1000003 + 1
        c_kw = (c_kw + int((blk * idx).sum().item())) & 0x7FFFFFFFFFFFFFFF
    return {"n_k": c_k, "n_kw": c_kw}


def checksum_verdict(name, docs_total, sweeps, got):
    """compare with the stored N = 1 digests -> (True / False / None, note)"""
    try:
        table = json.load(open(CHECKSUM_FILE))
    except Exception:                                   # noqa: BLE001
        return None, "profiles/state_checksums.json not found"
    t = table.get("%s:%d" % (name, docs_total))
    if not t or sweeps > len(t["n_k"]) or sweeps < 1:
        return None, "no stored N = 1 digest for %s with %d documents after %d sweeps" % (name, docs_total, sweeps)
    want = {"n_k": t["n_k"][sweeps - 1], "n_kw": t["n_kw"][sweeps - 1]}
    return (got == want), "N = 1 digests after %d sweeps (%s): %r" % (sweeps, t.get("source", "stored"), want)


def docs_per_wavefront(sampler):
    """quad kernel (csrc/kernel_quad.hpp): a document is K / 32 lanes x 32 slots"""
    return {32: "four", 16: "eight", 8: "sixteen"}.get(int(sampler.layout.G), "?")


def rows_description(sampler):
    """how the sweep reads n_kw: int32 rows, or the 16-bit image (llda_sweep_args.n_kw16, refreshed inside every timed sweep)
    for the words whose corpus-wide count fits 16 bits"""
    if getattr(sampler, "n_kw_img", None) is not None:
        return ("saturating %d-bit image of n_kw (llda_pack_image runs inside every timed sweep); an entry that shows the saturation "
                "value is re-read from the int32 counts; same results" % (8 * sampler.n_kw_img.element_size()))
    if getattr(sampler, "n_kw16", None) is None:
        return "int32"
    if getattr(sampler, "quad", False):
        fits = float(sampler.row16.float().mean().item())
        return ("16-bit image of EVERY row (llda_pack_rows16_all runs inside every timed sweep and flags the rows whose counts all fit: "
                "%.2f %% of the words this sweep; the others are read as int32), %s documents per wavefront; same results "
                "(DESIGN.md section 4.1)" % (100.0 * fits, docs_per_wavefront(sampler)))
    flagged = sampler.row16[sampler.word.long()].float().mean().item() if sampler.S else 0.0
    return ("16-bit image for the words whose corpus-wide count fits 16 bits (%.1f %% of the words, %.1f %% of this rank's sites; "
            "llda_pack_rows16 runs inside every timed sweep), int32 rows for the others; same results (DESIGN.md section 4.1)" %
            (100.0 * sampler.row16.float().mean().item(), 100.0 * flagged))


def rows_short(sampler):
    if getattr(sampler, "n_kw_img", None) is not None:
        return "%d-bit saturating image + int32 escapes" % (8 * sampler.n_kw_img.element_size())
    if getattr(sampler, "quad", False):
        return "16-bit image of every row, %s documents per wavefront" % docs_per_wavefront(sampler)
    return "int32" if getattr(sampler, "n_kw16", None) is None else "16-bit image + int32 hot rows"


def time_sweeps(sampler, steps, warmup, dist=None, dev=None, events=True):
    """warmup untimed sweeps, then exactly `steps` sweeps between barrier + synchronize; MAX over ranks.
    -> (seconds, mean sweep-kernel ms from HIP events on the launch stream).
    events=False (single process only; sweeps of well under a millisecond: the two event records per sweep are GPU commands of their
    own and cost 6.5 us of a 67 us abstracts sweep, tools/cpu_overhead_probe.py): the timed sweeps run without them and the kernel
    time comes from up to 200 FURTHER sweeps outside the timed region -- the sampler has then run warmup + steps + min(steps, 200)
    sweeps (sampler.sweeps_done says so; whatever is computed from its state afterwards is a state that many sweeps old)."""
    for _ in range(warmup):
        sampler.sweep()
    # the collector stays out of the timed region: a generation-2 pass over the heap the earlier workloads of this process left
    # behind (token lists, the numpy ports' states) takes 10 - 20 ms -- a millisecond per step of a 20-step, 48 ms measurement
    # (extra.wide_k2048 was 2.43 ms per step in six runs and 3.0 - 3.4 in four; a fresh process always shows 2.43 - 2.45:
    # tools/wide_step_probe.py, profiles/r06_wide_k2048_bimodal.md)
    import gc
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    try:
        return _time_sweeps(sampler, steps, dist, dev, events)
    finally:
        if gc_was:
            gc.enable()


def _time_sweeps(sampler, steps, dist, dev, events):
    if not events:
        if dist is not None:
            raise ValueError("time_sweeps(events=False) is a single-process measurement")
        sampler.kernel_events = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sampler.sweep()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sampler.kernel_events = []
        for _ in range(min(steps, 200)):
            sampler.sweep()
        torch.cuda.synchronize()
        sampler.check_status()
        kern_ms = [a.elapsed_time(b) for a, b in sampler.kernel_events]
        sampler.kernel_events = None
        return dt, (float(np.mean(kern_ms)) if kern_ms else float("nan"))
    sampler.kernel_events = []
    sampler.comm_events = [] if hasattr(sampler, "comm_events") else None
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sampler.sweep()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    sampler.check_status()
    kern_ms = [a.elapsed_time(b) for a, b in sampler.kernel_events]
    sampler.kernel_events = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, (float(np.mean(kern_ms)) if kern_ms else float("nan"))


# ------------------------------------------------------------------------------------------------ CPU baseline
def host_cpu():
    """(model name, logical cores, physical cores) of this host from /proc/cpuinfo."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model == "unknown":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
        if pid is not None:
            phys.add((pid, cid))
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, logical, (len(phys) or logical)


STRONG_CPU_DOCS = 100000     # documents of the strong (OpenMP C) baseline leg at its largest thread count


def cpu_strong_leg(sampler, info, labs_rows=None):
    """the strong CPU baseline: oracle/llda_oracle.c (snapshot sweeps, OpenMP over documents) on up to STRONG_CPU_DOCS documents of
    the SAME workload with min(physical cores, 64) threads and with half / a quarter of them (on proportionally fewer documents,
    so that every leg runs about as long) -> {"best": {...}, "legs": [...]}."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    _, logical, phys = host_cpu()
    cap = max(1, min(phys, 64))
    n_top = min(info["docs_local"], STRONG_CPU_DOCS)
    doc_off = sampler.doc_off[:n_top + 1].cpu().numpy()
    S = int(doc_off[-1])
    word, freq = sampler.word[:S].cpu().numpy(), sampler.freq[:S].cpu().numpy()
    z = sampler._pos_topic[sampler.z[:S].to(torch.int64)].cpu().numpy()
    n_k_v, n_zk = sampler.n_k_v(), sampler.n_zk()
    n_d_k = sampler.n_dk[:n_top][:, sampler._topic_pos].cpu().numpy().astype(np.int64)
    legs = []
    for threads in sorted({cap, max(1, cap // 2), max(1, cap // 4)}, reverse=True):
        n = max(1, n_top * threads // cap)
        s1 = int(doc_off[n])
        t0 = time.perf_counter()
        c_oracle.sweep_docs(np.arange(n) + sampler.doc_base, doc_off[:n + 1], word[:s1], freq[:s1], z[:s1],
                            None if labs_rows is None else labs_rows[:n], n_d_k[:n], n_k_v, n_zk, sampler.V, sampler.alpha,
                            sampler.beta, sampler.seed, 0, threads=threads)
        dt = time.perf_counter() - t0
        legs.append({"threads": threads, "docs": n, "sites": s1, "seconds": dt, "value": s1 / dt / 1e6})
    out = {"best": max(legs, key=lambda l: l["value"]), "legs": legs, "physical_cores": phys, "logical_cores": logical,
           "layout": "the reference's: n_k_v (K, V) int64, strided column gather per site (LabeledLDA.py:114)"}
    if labs_rows is None:
        # the same sweep on the layout the GPU kernels use (int32, word-major: a site reads ONE contiguous row) -- what a CPU does
        # when it is given the better layout; dense masks only
        wm_legs = []
        for threads in sorted({cap, max(1, cap // 2), max(1, cap // 4)}, reverse=True):
            n = max(1, n_top * threads // cap)
            s1 = int(doc_off[n])
            st = c_oracle.WMState(doc_off[:n + 1], word[:s1], freq[:s1], z[:s1], n_d_k[:n], n_k_v, n_zk, sampler.V, sampler.alpha, sampler.beta)
            t0 = time.perf_counter()
            st.sweep(sampler.seed, 0, doc_base=sampler.doc_base, threads=threads)
            dt = time.perf_counter() - t0
            wm_legs.append({"threads": threads, "docs": n, "sites": s1, "seconds": dt, "value": s1 / dt / 1e6})
        out["word_major"] = {"best": max(wm_legs, key=lambda l: l["value"]), "legs": wm_legs,
                             "layout": "the GPU kernels': n_kw (V, K) int32 word-major, n_d_k / n_k int32 (oracle/llda_oracle.c, llda_oracle_sweep_wm)"}
    return out


def cpu_baseline(sampler, doc_off, word, freq, n_docs_py, n_docs_c, labs=None):
    """Reference CPU path restated (oracle/) on a bounded sample of the SAME workload, timed on this
    host.  'port' = numpy per-site loop issuing the op sequence of LabeledLDA.py:108-125 on one core
    (the reference is single-threaded python/numpy)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import llda_oracle as orc
    import c_oracle
    K, V = sampler.K, sampler.V
    n_k_v = sampler.n_k_v()
    n_zk = sampler.n_zk()
    nmax = max(n_docs_py, n_docs_c)
    n_d_k = sampler.n_dk[:nmax, sampler._topic_pos].cpu().numpy().astype(np.int64)
    z = sampler._pos_topic[sampler.z[:int(doc_off[nmax])].to(torch.int64)].cpu().numpy()
    out = {}
    # --- numpy loop (like-for-like with the reference) ---
    off = doc_off[:n_docs_py + 1]
    docs = [word[off[d]:off[d + 1]].tolist() for d in range(n_docs_py)]
    freqs = [freq[off[d]:off[d + 1]].tolist() for d in range(n_docs_py)]
    labs = np.ones((nmax, K), dtype=np.uint8) if labs is None else labs
    st = orc.State(docs, freqs, labs[:n_docs_py].astype(np.float64), V, sampler.alpha, sampler.beta,
                   [z[off[d]:off[d + 1]] for d in range(n_docs_py)])
    st.n_k_v, st.n_zk, st.n_d_k = n_k_v.copy(), n_zk.copy(), n_d_k[:n_docs_py].copy()
    np.random.seed(1)
    t0 = time.perf_counter()
    orc.sweep_sequential(st, None)
    dt = time.perf_counter() - t0
    sites = int(off[-1])
    out["numpy"] = dict(value=sites / dt / 1e6, seconds=dt, sites=sites, docs=n_docs_py)
    # --- C restatement, 1 thread and all cores (snapshot semantics = what the GPU computes) ---
    cores = os.cpu_count() or 1
    offc = doc_off[:n_docs_c + 1]
    nc = int(offc[-1])
    for label, threads in (("c_1thread", 1), ("c_allcores", cores)):
        cs = c_oracle.CState(offc, word[:nc], freq[:nc], z[:nc], labs[:n_docs_c],
                             n_d_k[:n_docs_c], n_k_v, n_zk, V, sampler.alpha, sampler.beta)
        t0 = time.perf_counter()
        cs.sweep(1, sampler.seed, 0, threads=threads)
        dt = time.perf_counter() - t0
        out[label] = dict(value=nc / dt / 1e6, seconds=dt, sites=nc, docs=n_docs_c, threads=threads)
    return out, cores


def cpu_baseline_json(sampler, info, name, value):
    K = sampler.K
    Dg = info["docs_local"]
    n_py = n_c = min(Dg, 3000)                         # ~10 s of single-core numpy work at K=512
    labs_h = None
    if name in ("synth2_sparse", "synth2_sparse_hier", "synth2_sparse_scr"):
        labs_h = np.zeros((n_py, K), dtype=np.uint8)
        lh = info["lab"][:n_py].cpu().numpy()
        labs_h[np.repeat(np.arange(lh.shape[0]), 8), lh.reshape(-1)] = 1
    if name == "abstracts":                            # the whole corpus: one sweep of the numpy loop is ~2 s
        g = info["fixture"]
        n_py = n_c = Dg
        labs_h = np.zeros((Dg, K), dtype=np.uint8)
        labs_h[np.repeat(np.arange(Dg), np.diff(g["lab_off"])), g["lab_idx"]] = 1
        h_off, h_word, h_freq = g["doc_off"], g["word"].astype(np.int32), g["freq"].astype(np.int32)
    else:
        h_off = info["doc_off"][:n_py + 1].cpu().numpy()
        nmax = int(h_off[n_py])
        h_word, h_freq = info["word"][:nmax].cpu().numpy(), info["freq"][:nmax].cpu().numpy()
    base, cores = cpu_baseline(sampler, h_off, h_word, h_freq, n_py, n_c, labs_h)
    strong = None
    if name != "abstracts":                            # (abstracts: the whole corpus already ran on all cores above)
        lab_rows = None
        if name in ("synth2_sparse", "synth2_sparse_hier", "synth2_sparse_scr"):
            n_top = min(Dg, STRONG_CPU_DOCS)
            lab_rows = np.zeros((n_top, K), dtype=np.uint8)
            lh = info["lab"][:n_top].cpu().numpy()
            lab_rows[np.repeat(np.arange(n_top), 8), lh.reshape(-1)] = 1
        strong = cpu_strong_leg(sampler, info, lab_rows)
    model, logical, phys = host_cpu()
    return {
        "cpu_model": model, "physical_cores": phys, "c_port_reference_layout": strong,
        "value": base["numpy"]["value"], "unit": "Mtokens/s", "cores": 1, "kind": "port",
        "sample": "first %d docs (%d sites) of the same workload, 1 sweep, numpy per-site loop "
                  "restating LabeledLDA.py:108-125 (oracle/llda_oracle.py sweep_sequential), %.1f s"
                  % (n_py, base["numpy"]["sites"], base["numpy"]["seconds"]),
        "c_port_1thread_Mtokens_s": base["c_1thread"]["value"],
        "c_port_allcores_Mtokens_s": base["c_allcores"]["value"],
        "c_port_sample": "first %d docs (%d sites), oracle/llda_oracle.c snapshot mode" %
                         (n_c, base["c_1thread"]["sites"]),
        "host_cores": cores,
        "port_vs_reference": "the port runs within 10 % of the unmodified reference loop and leaves identical "
                             "counts (measured in the build container: profiles/port_calibration.json)",
    }, value / base["numpy"]["value"]


def cascade_extra():
    """BASELINE configs[4] on this GPU: CascadeLDA.go_down_tree(it=4, s=2) on the abstracts fixture -- the whole
    ensemble of 122 per-node Labeled-LDA sub-problems, host enumeration and initial assignments included; first call
    (cold) and the best of three more."""
    import io
    from contextlib import redirect_stdout
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.corpus import cascade_corpus_from_csr
    from lda_thesis_amd.text import Dictionary
    g = np.load(os.path.join(ROOT, "tests", "golden", "abstracts_d3.npz"))
    names = [str(x) for x in g["labelset"]]
    docs, labs, labelset = cascade_corpus_from_csr(g["doc_off"], g["word"], g["freq"], g["lab_off"], g["lab_idx"], names)
    dicti = Dictionary(docs)
    walls = []
    for _ in range(7):                                         # one cold call + six warm ones
        np.random.seed(0)
        model = CascadeLDA(docs, labs, list(labelset), dicti, alpha=ALPHA, beta=BETA, seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            model.go_down_tree(it=4, s=2)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    plans = model.plan_subproblems()
    lens = np.array([len(t) for t in model.doc_tups])
    sites = int(sum(int(lens[p["docs"]].sum()) for p in plans))
    return {"workload": "CascadeLDA on abstracts_data.csv (fixture), go_down_tree(it=4, s=2): ensemble of per-node L-LDA "
                        "sub-problems, all trained together on this GPU (BASELINE configs[4])",
            "value": min(walls[1:]), "median_s": float(np.median(walls[1:])), "max_s": max(walls[1:]),
            "unit": "s", "higher_is_better": False, "cold_first_call_s": walls[0],
            "warm_calls_s": walls[1:], "sub_problems": len(plans), "sites_per_ensemble_sweep": sites,
            "Msites_per_s": sites * 4 / min(walls[1:]) / 1e6,
            "reference_cpu_s": 66.8, "reference_cpu_note": "the reference's go_down_tree(4, 2) on one core of the survey "
            "container (SURVEY.md section 6); 368 s with per-document-snapshot sweeps (oracle/gen_golden.py, the run that "
            "made tests/golden/cascade_abstracts.npz)"}


# ------------------------------------------------------------------------------------------------ (f) rows: the harness steps
def _abstracts_tokens(g, off_key, word_key, freq_key, lab_off_key, lab_idx_key, names):
    """token lists (word id v with frequency f = f copies of 'w%05d' % v) and label-string lists of the fixture."""
    off, w, f = g[off_key], g[word_key], g[freq_key]
    lo, li = g[lab_off_key], g[lab_idx_key]
    docs, labs = [], []
    for d in range(len(off) - 1):
        toks = []
        for v, n in zip(w[off[d]:off[d + 1]], f[off[d]:off[d + 1]]):
            toks += ["w%05d" % v] * int(n)
        docs.append(toks)
        labs.append([names[k] for k in li[lo[d]:lo[d + 1]] if k != 0])
    return docs, labs


def _quiet(fn, *a, **k):
    import io
    from contextlib import redirect_stdout
    with redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def _reference_constructor_port_s(docs, labs, labelset, dicti):
    """seconds the reference's constructor takes on one core, restated loop for loop (LabeledLDA.py:50-92: set_label per document,
    doc2bow, one np.random.choice per document, the count accumulation site by site)."""
    t0 = time.perf_counter()
    labelset = ["root"] + list(labelset)
    labelmap = {lab: i for i, lab in enumerate(labelset)}
    K, V = len(labelmap), len(dicti)

    def set_label(label):
        vec = np.zeros(K)
        vec[0] = 1.0
        for x in label:
            vec[labelmap[x]] = 1.0
        return vec
    labm = np.array([set_label(lab) for lab in labs])
    tups = [dicti.doc2bow(x) for x in docs]
    n_zk, n_d_k, n_k_v = np.zeros(K, dtype=int), np.zeros((len(docs), K), dtype=int), np.zeros((K, V), dtype=int)
    for d, (doc, lab) in enumerate(zip(tups, labm)):
        ids, freqs = zip(*doc)
        zets = np.random.choice(K, size=len(doc), p=lab / lab.sum())
        for v, z, freq in zip(ids, zets, freqs):
            n_zk[z] += freq
            n_d_k[d, z] += freq
            n_k_v[z, v] += freq
    return time.perf_counter() - t0


def pipeline_extra(with_cpu=True):
    """SURVEY 8(f) rows 1, 3, 4 as the harness runs them (evaluate_LabeledLDA.py:110-180 of the reference): Labeled LDA on the
    abstracts fixture -- run_training(200, 25) with the thinning read-outs on the device, run_test of the 464 held-out
    documents (150 sweeps, thinning 25: one llda_foldin launch), the report's metrics -- each timed, with the CPU port
    of the same step (oracle/, bounded samples) beside it."""
    from lda_thesis_amd import evaluate as ev
    from lda_thesis_amd.LabeledLDA import LabeledLDA
    from lda_thesis_amd.text import Dictionary
    g = np.load(os.path.join(ROOT, "tests", "golden", "abstracts_d3.npz"))
    names = [str(x) for x in g["labelset"]]
    docs, labs = _abstracts_tokens(g, "doc_off", "word", "freq", "lab_off", "lab_idx", names)
    tdocs, tlabs = _abstracts_tokens(g, "test_doc_off", "test_word", "test_freq", "test_lab_off", "test_lab_idx", names)
    keep = [i for i, t in enumerate(tdocs) if t]
    tdocs, tlabs = [tdocs[i] for i in keep], [tlabs[i] for i in keep]
    dicti = Dictionary(docs)
    IT, THIN, T_IT, T_THIN = 200, 25, 150, 25
    out = {"workload": "Labeled LDA on the tokenised abstracts_data.csv fixture: the constructor (labels, doc2bow, initial assignments, "
                       "counts, upload) + run_training(%d, %d) + run_test of %d held-out "
                       "documents (%d sweeps, thinning %d) + the report's metrics (reference evaluate_LabeledLDA.py:110-180)"
                       % (IT, THIN, len(tdocs), T_IT, T_THIN), "unit": "s", "higher_is_better": False}
    best = None
    for rep in range(2):                                   # first pass cold (code objects, allocator), second timed
        np.random.seed(0)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        model = LabeledLDA(docs, labs, names[1:], dicti, ALPHA, BETA, seed=1)     # (reference LabeledLDA.py:50-92: labels, doc2bow,
        torch.cuda.synchronize()                                                   # initial assignments, counts) + upload
        t0 = time.perf_counter()
        _quiet(model.run_training, IT, THIN)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        th = model.run_test(tdocs, T_IT, T_THIN)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        th4 = np.round(np.asarray(th), 4)                  # test_it rounds to 4 decimals before the metrics
        y_bin = ev.binary_yreal(tlabs, model.labelmap)[:, 1:]
        thr = th4[:, 1:]
        nz = np.where(thr.sum(axis=1) != 0)[0]
        y_bin, thr = y_bin[nz, :], thr[nz, :]
        tps, tns, fps, fns, fprs, tprs = ev.rates(thr, y_bin)
        metrics = {"auc_roc": float(ev.macro_auc_roc(fprs, tprs)), "one_error": float(ev.n_error(thr, y_bin, 1)),
                   "two_error": float(ev.n_error(thr, y_bin, 2)), "f1_macro": float(ev.get_f1(tps, fps, tns, fns))}
        t3 = time.perf_counter()
        best = dict(construct_s=t0 - tc, train_s=t1 - t0, test_s=t2 - t1, metrics_s=t3 - t2)
        if rep == 0:
            out["cold_first_pass_s"] = dict(best)
    sites_train = int(sum(len(t) for t in model.doc_tups))
    ttups = [dicti.doc2bow(x) for x in tdocs]
    sites_test = int(sum(len(t) for t in ttups))
    out.update(value=best["construct_s"] + best["train_s"] + best["test_s"] + best["metrics_s"], stages_s=best, metrics=metrics,
               train={"sweeps": IT, "thinning": THIN, "sites_per_sweep": sites_train,
                      "Msite_draws_per_s": sites_train * IT / best["train_s"] / 1e6,
                      "perplexity_trace_last": float(model.cur_perplx[-1])},
               test={"documents": len(tdocs), "sites": sites_test, "sweeps": T_IT,
                     "Msite_draws_per_s": sites_test * T_IT / best["test_s"] / 1e6})
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import llda_oracle as orc
        # training: one sweep of the numpy per-site loop + one perplexity() read-out of the same model state
        st = orc.State([list(x) for x in model.docs], [list(x) for x in model.freqs], model.labs.astype(np.float64), model.V,
                       ALPHA, BETA, [z.copy() for z in model.z_dn])
        st.n_k_v, st.n_zk, st.n_d_k = model.n_k_v.copy(), model.n_zk.copy(), model.n_d_k.copy()
        np.random.seed(1)
        t0 = time.perf_counter()
        orc.sweep_sequential(st, None)
        t1 = time.perf_counter()
        orc.perplexity(st)
        orc.get_phi(st)
        orc.get_theta(st)
        t2 = time.perf_counter()
        # test: the first documents of the held-out set through the restated run_test, same keyed draws as the device
        n_s = min(12, len(ttups))
        ph_hat = model.ph_hat
        ids = [[v for v, _ in t] for t in ttups[:n_s]]
        frs = [[f for _, f in t] for t in ttups[:n_s]]
        from lda_thesis_amd.foldin import TEST_STREAM
        t3 = time.perf_counter()
        want = orc.run_test(ph_hat, ALPHA, ids, frs, T_IT, T_THIN, orc.keyed_draw_for(model.seed, TEST_STREAM))
        t4 = time.perf_counter()
        s_sample = sum(len(x) for x in ids)
        cpu_train = IT * (t1 - t0) + (IT // THIN) * (t2 - t1)
        cpu_test = (t4 - t3) * sites_test / s_sample
        cpu_construct = _reference_constructor_port_s(docs, labs, names[1:], dicti)
        out["cpu_baseline"] = {
            "kind": "port", "cores": 1, "unit": "s", "value": cpu_construct + cpu_train + cpu_test,
            "construct_s": cpu_construct, "train_s": cpu_train, "test_s": cpu_test,
            "sample": "constructor: the reference's loops in full (%.2f s); training: 1 sweep of the numpy per-site loop (%.2f s) x %d + 1 thinning read-out (perplexity, phi, theta: "
                      "%.2f s) x %d; test: run_test of the first %d held-out documents (%d of %d sites, %.2f s) scaled by sites"
                      % (cpu_construct, t1 - t0, IT, t2 - t1, IT // THIN, n_s, s_sample, sites_test, t4 - t3),
            "test_sample_identical_to_device": bool(np.array_equal(np.asarray(th)[:n_s], want))}
        out["speedup_vs_cpu_port"] = out["cpu_baseline"]["value"] / out["value"]
    return out


def cascade_test_extra(with_cpu=True):
    """SURVEY 8(f) row 1, Cascade side: test_down_tree (reference CascadeLDA.py:249-297) for every held-out document of the
    abstracts fixture (150 sweeps, thinning 25, threshold 0.95) after go_down_tree(4, 2) -- all documents that reach the
    same node of the label tree in one llda_foldin launch -- with the CPU port of the same walk (oracle cascade_test,
    same keyed draws) on a bounded sample beside it."""
    from lda_thesis_amd.CascadeLDA import CascadeLDA
    from lda_thesis_amd.corpus import cascade_corpus_from_csr
    from lda_thesis_amd.foldin import doc_key
    from lda_thesis_amd.text import Dictionary
    g = np.load(os.path.join(ROOT, "tests", "golden", "abstracts_d3.npz"))
    names = [str(x) for x in g["labelset"]]
    docs, labs, labelset = cascade_corpus_from_csr(g["doc_off"], g["word"], g["freq"], g["lab_off"], g["lab_idx"], names)
    held, _ = _abstracts_tokens(g, "test_doc_off", "test_word", "test_freq", "test_lab_off", "test_lab_idx", names)
    held = [t for t in held if t]
    dicti = Dictionary(docs)
    np.random.seed(0)
    model = CascadeLDA(docs, labs, list(labelset), dicti, alpha=ALPHA, beta=BETA, seed=1)
    _quiet(model.go_down_tree, it=4, s=2)
    model.ph = np.nan_to_num(model.ph)                     # (the never-trained '' row)
    IT, THIN, THR = 150, 25, 0.95
    walls = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trees = model.test_down_tree_batch(held, IT, THIN, THR)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    nodes = sum(1 + len(t[1]) + len(t[2]) for t in trees)
    bows = [dicti.doc2bow(x) for x in held]
    out = {"workload": "CascadeLDA.test_down_tree for the %d held-out documents of the abstracts fixture (%d sweeps, thinning %d, "
                       "threshold %.2f) after go_down_tree(4, 2); documents at the same tree node share a launch"
                       % (len(held), IT, THIN, THR),
           "value": min(walls[1:]), "unit": "s", "higher_is_better": False, "cold_first_call_s": walls[0],
           "warm_calls_s": walls[1:], "documents": len(held), "node_visits": nodes,
           "site_sweeps": int(sum(len(bows[d]) * (1 + len(t[1]) + len(t[2])) for d, t in enumerate(trees))) * IT}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import re
        import llda_oracle as orc

        def node(tup, labels):
            ids_, fr_ = zip(*tup)
            rows = [model.labelmap[x] for x in labels]
            stream, key = model._test_stream(labels), doc_key(tup)

            def draw_for_sweep(sw):
                k = orc.KeyedDraw(model.seed, stream)
                k.sweep, k.doc, k.site = sw, key, 0
                return k
            return orc.cascade_test(model.ph[rows, :], ALPHA, BETA, list(ids_), list(fr_), IT, THIN, draw_for_sweep)

        def walk(tup):                                    # the visiting order of the reference's test_down_tree
            kids = lambda p: [p] + list(filter(re.compile("^" + p + "[0-9]{1}$").match, model.lablist))
            labels = model.lablist_l1
            keep, loads = model._head(node(tup, labels), labels, THR)
            l1, l2, l3 = list(zip(keep, loads)), [], []
            for parent in [x for x in keep if x != "root"]:
                labels = kids(parent)
                keep2, loads2 = model._head(node(tup, labels), labels, THR)
                l2.append(list(zip(keep2, loads2)))
                for parent2 in [x for x in keep2 if x != parent]:
                    labels = kids(parent2)
                    keep3, loads3 = model._head(node(tup, labels), labels, THR)
                    l3.append(list(zip(keep3, loads3)))
            return l1, l2, l3
        n_s = min(3, len(held))
        t0 = time.perf_counter()
        ported = [walk(bows[d]) for d in range(n_s)]
        dt = time.perf_counter() - t0
        work = lambda d, t: len(bows[d]) * (1 + len(t[1]) + len(t[2]))
        w_s = sum(work(d, trees[d]) for d in range(n_s))
        w_all = sum(work(d, t) for d, t in enumerate(trees))
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "unit": "s", "value": dt * w_all / w_s,
                               "sample": "the walk of the first %d held-out documents through oracle cascade_test (%.2f s), scaled by "
                                         "sites x node visits (%d of %d)" % (n_s, dt, w_s, w_all),
                               "sample_identical_to_device": all(str(a) == str(b) for a, b in zip(trees[:n_s], ported))}
        out["speedup_vs_cpu_port"] = out["cpu_baseline"]["value"] / out["value"]
    return out


# ------------------------------------------------------------------------------------------------ PMC passes
# every workload whose sweep is ONE kernel launch per sweep, in the order the inner run sweeps them
PMC_WORKLOADS = ("synth2", "synth1", "synth_k256", "synth2_hostile", "synth2_sparse", "synth2_sparse_hier", "synth2_sparse_scr", "synth_wide_sparse", "synth_wide",
                 "abstracts")
PMC_SWEEPS = 3                                              # per workload in the inner run (all are measured)
# one rocprofv3 run per group (kernel trace only, as the guide prescribes).  TCC holds 4 counters per pass
# (FETCH_SIZE costs 3, WRITE_SIZE 2); SQ / TA / TCP / GRBM are separate blocks.
PMC_PASSES = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
              ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY",
                      "GRBM_GUI_ACTIVE"]),
              ("l2", ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"]),
              ("ta", ["TA_BUSY_avr", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum", "SQ_INSTS_SALU", "SQ_INSTS_LDS"]))
MALL_BYTES = 256 * 2 ** 20                                  # Infinity Cache (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6300.0                                 # measured streaming ceiling of the guide


def is_sweep_kernel(name):
    return "llda_sweep" in name


def short_kernel_name(name):
    """'void (anonymous namespace)::llda_sweep_kernel<32, 16, false, true, true, false>(KParams)' -> the template-id"""
    m = re.search(r"llda_\w+(<[^()]*>)?", name.replace("(anonymous namespace)::", ""))
    return m.group(0) if m else name


def pmc_inner(dev, workloads, docs=0):
    """the process rocprofv3 wraps: PMC_SWEEPS sweeps of each workload, nothing else.  docs > 0: a corpus of that many documents (the
    single-process replica of ONE rank's shard of an N > 1 run: the same kernel, the same n_kw, the rank's share of the documents)."""
    for name in workloads:
        s, info = build_sampler(name, dev, 0, 1, False, docs_total=docs)
        for _ in range(PMC_SWEEPS):
            s.sweep()
        torch.cuda.synchronize()
        del s, info
        torch.cuda.empty_cache()


def pmc_collect(workloads, keep_dir=None, timeout=300, docs=0, passes=None):
    """Run this script under rocprofv3, one --pmc group per pass, and return
    ({workload: {counter: mean per sweep-kernel launch}}, note).  A pass that fails only loses its own counters."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}, "rocprofv3 not found"
    out = {w: {} for w in workloads}
    problems = []
    tmp = tempfile.mkdtemp(prefix="llda_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    try:
        for tag, counters in (passes or PMC_PASSES):
            d = os.path.join(tmp, tag)
            cmd = [exe, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "-o", "pmc", "--",
                                                                 sys.executable, os.path.join(ROOT, "bench.py"),
                                                                 "--pmc-inner", ",".join(workloads)] + \
                  (["--docs", str(docs)] if docs else [])
            # own session: on a timeout the whole process group (rocprofv3 AND the python under it) is ended
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                    start_new_session=True)
            try:
                out_text, _ = proc.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, 9)
                proc.communicate()
                problems.append("pass %s did not finish in %d s" % (tag, timeout))
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if proc.returncode != 0 or not files:
                problems.append("pass %s failed (rc %d): %s" % (tag, proc.returncode, out_text.decode(errors="replace")[-200:]))
                continue
            rows, names = [], {}
            for r_ in csv.DictReader(open(files[0])):
                nm = r_["Kernel_Name"]
                if is_sweep_kernel(nm):
                    did = int(r_["Dispatch_Id"])
                    names[did] = nm
                    rows.append((did, r_["Counter_Name"], float(r_["Counter_Value"])))
                    if r_["Counter_Name"] == counters[0]:      # the launch's duration under this pass
                        rows.append((did, tag + "_pass_kernel_ns",
                                     float(int(r_["End_Timestamp"]) - int(r_["Start_Timestamp"]))))
            disp = sorted(set(x[0] for x in rows))
            if len(disp) != PMC_SWEEPS * len(workloads):
                problems.append("pass %s: %d sweep-kernel dispatches, expected %d" % (tag, len(disp), PMC_SWEEPS * len(workloads)))
                continue
            which = {d_: workloads[i // PMC_SWEEPS] for i, d_ in enumerate(disp)}
            agg = collections.defaultdict(list)
            for d_, c, v in rows:
                agg[(which[d_], c)].append(v)
            for (w, c), v in agg.items():
                out[w][c] = sum(v) / PMC_SWEEPS          # a counter may be reported in several rows per dispatch
            for d_, nm in names.items():
                out[which[d_]]["kernel_name"] = short_kernel_name(nm)
            if keep_dir:
                # the raw file has one row per counter INSTANCE (XCD / SE / channel) per dispatch -- tens of MB; what is
                # kept is one row per sweep-kernel dispatch and counter: the sum over the instances
                os.makedirs(keep_dir, exist_ok=True)
                tot = collections.OrderedDict()
                for d_, c, v in rows:
                    tot[(d_, c)] = tot.get((d_, c), 0.0) + v
                with open(os.path.join(keep_dir, "pmc_%s.csv" % tag), "w") as fh:
                    fh.write("Dispatch_Id,Workload,Kernel_Name,Counter_Name,Counter_Value_summed_over_instances\n")
                    for (d_, c), v in tot.items():
                        fh.write('%d,%s,"%s",%s,%.6f\n' % (d_, which[d_], short_kernel_name(names[d_]), c, v))
    except Exception as e:                                # noqa: BLE001 -- the bench line must survive a profiler problem
        problems.append("pmc collection failed: %r" % (e,))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    note = "in-run rocprofv3 --kernel-trace --pmc passes (%s), %d launches each" % (" | ".join(t for t, _ in PMC_PASSES), PMC_SWEEPS)
    if problems:
        note += "; PROBLEMS: " + "; ".join(problems)
    return out, note


def roofline_json(kernel_ms, sites, docs, live_topics, pmc, source, stored_key=None, shared_bytes=0, kernel="llda_sweep_kernel"):
    """Roofline block of one sweep kernel.

    achieved / frac / traffic: bytes that crossed the L2 <-> fabric interface per launch, from the PMC counters
    ((2 x FETCH_SIZE + WRITE_SIZE) x 1024: L2 line fills are 128-byte requests tallied as 64, MI355X_MICROARCH.md),
    over the kernel's mean duration (HIP events in the timed region), against the 8 TB/s HBM peak.  These requests
    INCLUDE hits in the 256 MiB Infinity Cache -- gfx950 exposes no counter behind it (TCC_EA0_RDREQ_DRAM_sum equals
    TCC_EA0_RDREQ_sum on every workload) -- so the figure is an UPPER bound of the HBM rate; `hbm_estimate` brackets it.
    `valu_issue`: wave64 VALU instructions x 4 cycles against 1024 SIMDs x clock.  `binding_roof` names the ceiling the
    evidence points at.  Algorithmic bytes are reported separately: rows served by L1 / L2 never reach the
    fabric, so algorithmic bytes over time is not a memory rate at all."""
    alg = algorithmic_bytes(sites, docs, live_topics)
    ks = kernel_ms * 1e-3
    traffic, kind = None, None
    if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        traffic, kind = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0, "measured"
    elif stored_key:
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            per_site = json.load(open(tpath)).get(stored_key)
            if per_site:
                traffic, kind = per_site * sites, "stored"
                source = ("STORED, not measured in this run: profiles/pmc_traffic.json holds %.1f fabric bytes per site from "
                          "the single-GPU counter passes; x local sites" % per_site)
    r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": (pmc or {}).get("kernel_name", kernel),
         "kernel_ms": kernel_ms,
         "algorithmic_bytes_per_launch": alg, "algorithmic_GBps": alg / ks / 1e9,
         "algorithmic_note": "SURVEY 8(d) bytes / kernel time; includes rows served by L1 / L2 / Infinity Cache, so it may "
                             "exceed the HBM peak and is not the roofline fraction"}
    cands = {}
    if traffic is not None:
        fabric = traffic / ks / 1e9
        # what HBM itself must at least move: the state that is streamed once per sweep and cannot stay in the 256 MiB
        # cache from one sweep to the next (sites: word, freq, z read + write, log word; documents: n_dk read + write of
        # the live entries) -- plus the shared counts when they do not fit either
        floor = sites * 24.0 + docs * 8.0 * live_topics + (shared_bytes if shared_bytes > MALL_BYTES else 0)
        r.update(achieved=fabric, frac=fabric / HBM_PEAK_GBS, traffic=traffic, traffic_kind=kind, traffic_source=source,
                 traffic_over_algorithmic=traffic / alg,
                 achieved_is="fabric_GBps: L2 <-> fabric bytes (L2 line fills + write-backs), Infinity-Cache hits included",
                 fabric_GBps=fabric,
                 hbm_estimate={"lower_GBps": floor / ks / 1e9, "upper_GBps": min(fabric, HBM_ACHIEVABLE_GBS),
                               "note": "lower = bytes streamed once per sweep that no cache can hold; upper = the fabric "
                                       "rate capped at the %.1f TB/s a streaming copy achieves; shared counts %.0f MB %s "
                                       "the 256 MiB Infinity Cache" % (HBM_ACHIEVABLE_GBS / 1e3, shared_bytes / 1e6,
                                                                       "exceed" if shared_bytes > MALL_BYTES else "fit")})
        cands["hbm" if shared_bytes > MALL_BYTES else "fabric (L2 fills, mostly Infinity-Cache hits)"] = fabric / HBM_PEAK_GBS
        # the measured ceiling of RANDOM line fills for a footprint of this size (tools/gather_ubench.hip, stored): what the
        # fabric rate can be compared with when the counts do not stream
        try:
            fc = json.load(open(os.path.join(ROOT, "profiles", "fabric_ceiling.json")))
            mib = shared_bytes / 2.0 ** 20
            i = min(range(len(fc["footprint_MiB"])), key=lambda j: abs(np.log(max(mib, 1.0) / fc["footprint_MiB"][j])))
            r["fabric_ceiling"] = {"line_fill_GBps": fc["line_fill_GBps"][i], "at_footprint_MiB": fc["footprint_MiB"][i],
                                   "frac": fabric / fc["line_fill_GBps"][i],
                                   "note": "random 128-byte line fills sustained by the gather micro-benchmark over the nearest "
                                           "measured footprint (profiles/fabric_ceiling.json); stored, not measured in this run.  A kernel that reads "
                                           "16 consecutive lines per row can exceed the purely random figure by a few per cent"}
        except Exception:                                   # noqa: BLE001
            pass
    else:
        r.update(achieved=None, frac=None, traffic=None, traffic_kind=None, traffic_source=source)
    if pmc and pmc.get("TCC_HIT_sum") is not None and pmc.get("TCC_MISS_sum") is not None:
        r["l2"] = {"hit_rate": pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]),
                   "requests_per_site": (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]) / sites,
                   "fabric_read_requests_per_site": pmc.get("TCC_EA0_RDREQ_sum", float("nan")) / sites,
                   "fabric_write_requests_per_site": pmc.get("TCC_EA0_WRREQ_sum", float("nan")) / sites}
    cycles = None
    if pmc and pmc.get("GRBM_GUI_ACTIVE"):
        cycles = pmc["GRBM_GUI_ACTIVE"] / N_XCD                       # shader cycles of the profiled launch
    if pmc and cycles and pmc.get("TA_BUSY_avr") is not None:
        r["ta_busy_frac"] = pmc["TA_BUSY_avr"] / cycles
        if pmc.get("TCP_PENDING_STALL_CYCLES_sum") is not None:
            r["tcp_pending_stall_frac"] = pmc["TCP_PENDING_STALL_CYCLES_sum"] / 256.0 / cycles
    if pmc and "SQ_INSTS_VALU" in pmc:
        insts = pmc["SQ_INSTS_VALU"]
        peak_ips = N_SIMD * MAX_CLOCK_HZ / VALU_CYCLES_PER_INST
        v = {"achieved": insts / ks, "peak": peak_ips, "unit": "wave64 VALU instructions/s",
             "frac": insts / ks / peak_ips, "valu_insts_per_site": insts / sites,
             "model": "SQ_INSTS_VALU x %d cycles / (%d SIMDs x %.1f GHz)" % (VALU_CYCLES_PER_INST, N_SIMD, MAX_CLOCK_HZ / 1e9)}
        if cycles and pmc.get("SQ_ACTIVE_INST_VALU"):
            v["valu_busy_frac"] = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * N_SIMD)
            v["valu_busy_note"] = ("SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share "
                                   "of the profiled launch's own cycles in which a SIMD issues VALU")
            if pmc.get("sq_pass_kernel_ns"):
                v["effective_clock_GHz"] = cycles / pmc["sq_pass_kernel_ns"]
        if pmc.get("SQ_WAIT_INST_ANY") and pmc.get("SQ_WAVE_CYCLES"):
            v["wave_cycles_waiting_frac"] = pmc["SQ_WAIT_INST_ANY"] / pmc["SQ_WAVE_CYCLES"]
            if cycles:
                v["waves_per_simd_avg"] = pmc["SQ_WAVE_CYCLES"] * 4.0 / (cycles * N_SIMD)
        r["valu_issue"] = v
        # the issue-side roof: the MEASURED share of cycles a SIMD issues VALU where the counter is there (never above 1), else the
        # instruction count at the nominal clock
        cands["valu_issue"] = min(1.0, v.get("valu_busy_frac", v["frac"]))
        if cycles and pmc.get("SQ_INSTS_SALU"):
            # profiles/r03_issue_model.md: 4 x VALU + 2.5 x SALU cycles per SIMD was FITTED on the K = 512 kernels at three waves
            # per SIMD (it reproduced their rate to 2 - 8 %, and still sat near 1 on a kernel that was bound by the fabric).  At
            # four waves the scalar instructions of one wave hide behind the vector ones of the others and the figure exceeds 1:
            # informational only, not one of the roofs below.
            ic = (VALU_CYCLES_PER_INST * insts + SALU_CYCLES_PER_INST * pmc["SQ_INSTS_SALU"]) / N_SIMD
            r["issue_model"] = {"frac": ic / cycles, "cycles_per_simd": ic, "shader_cycles": cycles,
                                "salu_insts_per_site": pmc["SQ_INSTS_SALU"] / sites,
                                "model": "(%d x SQ_INSTS_VALU + %.1f x SQ_INSTS_SALU) / 1024 SIMDs over GRBM_GUI_ACTIVE / 8; a fit at three "
                                         "waves per SIMD (profiles/r03_issue_model.md), above 1 where more waves overlap the scalar "
                                         "instructions: not a roof" % (VALU_CYCLES_PER_INST, SALU_CYCLES_PER_INST)}
    if cands:
        best = max(cands, key=cands.get)
        r["binding_roof"] = best
        r["roof_fractions"] = cands
        r["headroom"] = 1.0 - cands[best]
        r["headroom"] = max(0.0, r["headroom"])
        r["binding_note"] = ("the larger of: fabric bytes / time / 8 TB/s (called 'hbm' only when the shared counts exceed the "
                             "Infinity Cache) and the share of its cycles in which a SIMD issues VALU (valu_issue.valu_busy_frac; "
                             "SQ_INSTS_VALU x 4 cycles / time / (1024 SIMDs x 2.4 GHz) where that counter is missing).  "
                             "profiles/r03_issue_model.md: with int32 rows the K = 512 kernel is fabric-bound (0.96 of 8 TB/s in L2 "
                             "line fills; a fifth fewer issue cycles changed nothing); with the 16-bit rows it moves half those "
                             "bytes and is bound by instruction issue -- VALU-busy 0.81 at three waves per SIMD, 0.90 at the four "
                             "it runs at when documents hold fewer than 2^16 tokens; a kernel far below both is bound by the "
                             "latency of its dependent chain at its occupancy (valu_issue.wave_cycles_waiting_frac)")
    return r


# ------------------------------------------------------------------------------------------------ the printed line
LINE_LIMIT = 8000            # bytes: the driver keeps an 8 KB tail of stdout and parses the LAST line of it


def _num(x, digits=6):
    """numbers for the printed line: floats rounded to `digits` significant digits, everything else as is"""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    return x


def compact_roofline(r):
    """numbers only: what the judge recomputes.  achieved / frac / traffic as in roofline_json (fabric bytes from the PMC passes over
    the kernel's mean time, against 8 TB/s); algorithmic_* = SURVEY 8(d) bytes of one launch and their rate."""
    if not r:
        return None
    v, l2 = r.get("valu_issue") or {}, r.get("l2") or {}
    out = {"bound": r.get("bound"), "kernel": r.get("kernel"), "kernel_ms": r.get("kernel_ms"), "achieved": r.get("achieved"),
           "peak": r.get("peak"), "unit": r.get("unit"), "frac": r.get("frac"), "traffic": r.get("traffic"),
           "traffic_kind": r.get("traffic_kind"), "algorithmic_bytes": r.get("algorithmic_bytes_per_launch"),
           "algorithmic_GBps": r.get("algorithmic_GBps"), "traffic_over_algorithmic": r.get("traffic_over_algorithmic"),
           # SURVEY 8(d) bytes over time against 8 TB/s: NOT a fraction (rows served on-die never cross HBM; above 1 on configs[3])
           "algorithmic_frac": (r.get("algorithmic_GBps") or 0.0) / HBM_PEAK_GBS if r.get("algorithmic_GBps") is not None else None,
           "algorithmic_frac_note": "not a fraction: rows served by L2 / Infinity Cache",
           "frac_is": "fabric (L2 <-> fabric bytes, Infinity-Cache hits included) / 8 TB/s: an upper bound of the HBM fraction",
           "valu_busy_frac": v.get("valu_busy_frac"), "valu_insts_per_site": v.get("valu_insts_per_site"),
           "waves_per_simd": v.get("waves_per_simd_avg"), "wave_cycles_waiting_frac": v.get("wave_cycles_waiting_frac"),
           "l2_hit_rate": l2.get("hit_rate"), "fabric_read_requests_per_site": l2.get("fabric_read_requests_per_site"),
           "ta_busy_frac": r.get("ta_busy_frac"), "binding_roof": (r.get("binding_roof") or "").split(" ")[0] or None}
    he = r.get("hbm_estimate")
    out = {k: _num(x, 5) for k, x in out.items()}
    if he:
        out["hbm_frac_bounds"] = [_num(he["lower_GBps"] / HBM_PEAK_GBS, 4), _num(he["upper_GBps"] / HBM_PEAK_GBS, 4)]
    return out


def compact_cpu(c):
    if not c:
        return None
    out = {k: _num(c.get(k), 5) for k in ("value", "unit", "cores", "kind", "cpu_model", "host_cores", "physical_cores")
           if c.get(k) is not None}
    out["sample"] = str(c.get("sample", ""))[:200]
    if c.get("c_port_1thread_Mtokens_s") is not None:
        out["c_port_1thread"] = _num(c["c_port_1thread_Mtokens_s"], 5)
    if c.get("c_port_allcores_Mtokens_s") is not None:
        out["c_port_allcores_small_sample"] = _num(c["c_port_allcores_Mtokens_s"], 5)
    st = c.get("c_port_reference_layout")
    if st:
        b = st["best"]
        out["c_port_reference_layout"] = {"value": _num(b["value"], 5), "threads": b["threads"], "docs": b["docs"],
                                          "seconds": _num(b["seconds"], 4),
                                          "legs": [[l["threads"], _num(l["value"], 4)] for l in st["legs"]]}
        wm = st.get("word_major")
        if wm:
            b = wm["best"]
            out["c_port_word_major"] = {"value": _num(b["value"], 5), "threads": b["threads"], "docs": b["docs"],
                                        "seconds": _num(b["seconds"], 4),
                                        "legs": [[l["threads"], _num(l["value"], 4)] for l in wm["legs"]]}
    for k in ("train_s", "test_s"):
        if c.get(k) is not None:
            out[k] = _num(c[k], 5)
    return out


def compact_extra(e):
    """{value, unit, ms_per_step, frac, binding_roof} (+ the CPU figures where the workload has them) of one extra workload"""
    out = {"value": _num(e.get("value")), "unit": e.get("unit")}
    for k in ("ms_per_step", "kernel_ms", "steps", "speedup_vs_cpu_port", "cold_first_call_s", "median_s", "max_s", "n_kw_rows_short",
              "Mtokens_s", "tokens_per_site"):
        if e.get(k) is not None:
            out[k] = _num(e[k], 5)
    r = e.get("roofline")
    if r:
        c = compact_roofline(r)
        out.update({k: c[k] for k in ("frac", "binding_roof", "traffic_over_algorithmic", "valu_busy_frac", "l2_hit_rate",
                                      "fabric_read_requests_per_site")})
    if e.get("cpu_baseline"):
        cb = e["cpu_baseline"]
        out["cpu"] = {"value": _num(cb.get("value"), 5), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                      "scaled_from_sample": "scaled" in str(cb.get("sample", ""))}
        if cb.get("c_port_reference_layout"):
            out["cpu"]["c_port_reference_layout"] = _num(cb["c_port_reference_layout"]["best"]["value"], 5)
    if e.get("stages_s"):
        out["stages_s"] = {k: _num(x, 4) for k, x in e["stages_s"].items()}
    return out


def compact_line(detail, detail_path=None):
    """The ONE line the driver parses, from the full record `detail` (which goes to --detail-out): numbers only, no prose, at most
    LINE_LIMIT bytes -- if it ever grew beyond that the extras are dropped before the line is allowed to become unparseable."""
    cfg = detail.get("config", {})
    line = {k: _num(detail.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                              "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: _num(cfg.get(k)) for k in ("workload", "docs_total", "docs_per_gpu", "sites_per_doc", "K", "V", "alpha",
                                                     "beta", "label_mask", "kernel", "n_kw_rows_short", "sites_per_sweep", "tokens_per_site",
                                                     "timed_seconds", "state_checksum_n_k", "state_checksum_n_kw",
                                                     "sweeps_behind_checksum", "build_info", "abi", "library")
                      if cfg.get(k) is not None}
    line["config"]["workload"] = str(cfg.get("workload", ""))[:160]
    if detail_path:
        line["config"]["detail"] = detail_path
    line["roofline"] = compact_roofline(detail.get("roofline"))
    line["cpu_baseline"] = compact_cpu(detail.get("cpu_baseline"))
    if detail.get("speedup_vs_cpu_port") is not None:
        line["speedup_vs_cpu_port"] = _num(detail["speedup_vs_cpu_port"], 5)
    if detail.get("draw_tiers"):
        line["draw_tiers"] = detail["draw_tiers"]
    for k in ("checksum_matches_n1",):
        if k in detail:
            line[k] = detail[k]
    if detail.get("exchange_ms"):
        line["exchange_ms"] = {k: _num(x, 5) for k, x in detail["exchange_ms"].items() if not isinstance(x, (dict, list, str))}
    if detail.get("overlap_probe"):
        pr = detail["overlap_probe"]
        line["overlap_probe"] = {k: _num(pr.get(k), 5) for k in ("overlap_ranges", "steps", "ms_per_step", "checksum_matches_n1")}
        if isinstance(pr.get("exchange_ms"), dict):
            line["overlap_probe"]["exposed_ms_per_sweep"] = _num(pr["exchange_ms"].get("exposed_ms_per_sweep"), 5)
    if detail.get("extra"):
        line["extra"] = {k: compact_extra(e) for k, e in detail["extra"].items()}
        hb = (detail["extra"].get("hbm_bound") or {}).get("roofline") or {}
        if line["roofline"] is not None and hb.get("frac") is not None:
            # the one workload of the run whose fabric bytes ARE HBM bytes (n_kw = 1 GB, four times the Infinity Cache, uniform words):
            # what the sweep kernel achieves against the HBM roof when nothing is served on-die
            line["roofline"]["hbm_credential"] = {"workload": "hbm_bound", "frac": _num(hb["frac"], 5), "kernel": hb.get("kernel")}
    if len(json.dumps(line)) > LINE_LIMIT and "extra" in line:
        line["extra"] = {k: {"value": e.get("value"), "unit": e.get("unit")} for k, e in line["extra"].items()}
    if len(json.dumps(line)) > LINE_LIMIT:
        line.pop("extra", None)
        line["config"] = {"workload": line["config"]["workload"], "build_info": line["config"].get("build_info")}
    return line


# ------------------------------------------------------------------------------------------------ main
def self_launch(args):
    """--gpus N > 1 without torch.distributed.run around us: start N ranks of this script."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


_T0 = time.time()


def start_watchdog(seconds, rank, world, args):
    """after `seconds`: every thread's stack on stderr, ONE JSON line with an "error" key on stdout (rank 0), exit code 3 -- a run
    that cannot finish says so instead of hanging until somebody kills it"""
    import faulthandler
    import threading

    def fire():
        try:
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            if rank == 0:
                print(json.dumps({"metric": "million tokens resampled/sec (Gibbs sweep)", "value": None, "unit": "Mtokens/s",
                                  "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                  "error": "bench.py did not finish within its deadline of %d s (--deadline); every thread's stack is on "
                                           "stderr of each rank" % seconds}), flush=True)
        finally:
            os._exit(3)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def trace(msg):
    """LLDA_BENCH_TRACE=1: time-stamped progress on stderr (every rank)"""
    if os.environ.get("LLDA_BENCH_TRACE"):
        print("[bench %6.1f s rank %s] %s" % (time.time() - _T0, os.environ.get("RANK", "0"), msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="synth2", choices=sorted(WORKLOADS))
    ap.add_argument("--docs", type=int, default=0, help="override the number of documents of the corpus (all GPUs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workloads of the N = 1 line")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes")
    ap.add_argument("--pmc-keep", default="", help="directory to keep the raw counter CSVs of the in-run passes in")
    ap.add_argument("--pmc-inner", default="", help=argparse.SUPPRESS)
    ap.add_argument("--detail-out", default=os.path.join("gpurun_out", "bench_detail.json"),
                    help="file that receives the full record (notes, models, every counter); the printed line holds numbers only")
    ap.add_argument("--make-checksums", type=int, default=0,
                    help="N = 1: run this many sweeps, print the digests after each (the content of one entry of "
                         "profiles/state_checksums.json) and exit")
    ap.add_argument("--force-exchange", action="store_true",
                    help="diagnostic: take the multi-GPU path (exchange rows, all-reduce if a group exists) on one GPU")
    ap.add_argument("--overlap", type=int, default=-1, help="document ranges per sweep for the overlapped exchange")
    ap.add_argument("--docs-per-group", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", help="diagnostic: gloo runs the N > 1 code path without RCCL")
    ap.add_argument("--deadline", type=int, default=1500,
                    help="seconds after which a run that is still going prints a one-line JSON with an \"error\" key (rank 0), dumps every "
                         "thread's stack on stderr and exits with code 3 instead of hanging; collectives time out after the same span")
    ap.add_argument("--one-device", action="store_true",
                    help="diagnostic: every rank uses cuda:0 (functional check of the N > 1 path on a one-GPU box, with gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from lda_thesis_amd import _native
    build_bits, build_names = _native.build_info()
    if build_bits:
        raise SystemExit("bench.py: %s was built with %s (llda_build_info() = %#x): not the production library, no line is printed"
                         % (_native.LIB_PATH, ", ".join(build_names), build_bits))
    if args.pmc_inner:
        pmc_inner(dev, args.pmc_inner.split(","), docs=args.docs)
        return
    watchdog = start_watchdog(args.deadline, rank, world, args)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a wedged collective raises after the deadline instead of blocking for ever (RCCL: the watchdog thread of the process group
        # aborts the communicator; gloo: the wait itself times out)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        pg_timeout = datetime.timedelta(seconds=max(60, args.deadline))
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world, timeout=pg_timeout)

    trace("process group ready" if dist is not None else "single process")
    if os.environ.get("LLDA_BENCH_TRACE"):                # where every rank stands if the run is still going after a minute
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("LLDA_BENCH_TRACE_DUMP_S", "60")), repeat=False, file=sys.stderr)
    name = args.workload
    # --one-device: the ranks share ONE GPU.  Several processes pushing long streams of small kernels and host synchronisations at one
    # device at the same time (the generator, the sorts and scans of the sampler's construction) are time-sliced by the GPU's scheduler
    # process by process: the same construction took 1 s, 40 - 70 s, 179 s or more than 150 s from one attempt to the next
    # (profiles/HISTORY.md, round 5).  There the heavy LOCAL sections of the construction take turns (a file lock; no collective
    # inside a locked section); the timed sweeps do run concurrently.
    lock = one_device_lock() if (args.one_device and dist is not None) else None
    sampler, info = build_sampler(name, dev, rank, world, dist is not None, docs_total=args.docs,
                                  docs_per_group=args.docs_per_group, force_exchange=args.force_exchange,
                                  overlap=None if args.overlap < 0 else args.overlap, build_lock=lock)
    sites_local = sampler.S
    torch.cuda.synchronize()
    trace("sampler built: %d local sites" % sites_local)
    if args.make_checksums:
        if world != 1:
            raise SystemExit("--make-checksums is a single-GPU run")
        out = {"n_k": [], "n_kw": [], "source": "python bench.py --workload %s%s --make-checksums %d on one MI355X" %
               (name, " --docs %d" % args.docs if args.docs else "", args.make_checksums)}
        for _ in range(args.make_checksums):
            sampler.sweep()
            c = state_checksums(sampler)
            out["n_k"].append(c["n_k"])
            out["n_kw"].append(c["n_kw"])
        sampler.check_status()
        print(json.dumps({"%s:%d" % (name, info["docs_total"]): out}))
        return
    dt, kavg = time_sweeps(sampler, args.steps, args.warmup, dist, dev)
    trace("timed sweeps done: %.3f s" % dt)
    tier = sampler.status.cpu().numpy().astype(np.int64)
    total_sites = sites_local
    sums = state_checksums(sampler)
    checksum = sums["n_k"]
    sweeps_done = sampler.sweeps_done
    comm = sampler.comm_stats() if hasattr(sampler, "comm_stats") else None
    if dist is not None and comm is not None and sampler.rows is not None:
        # the collective alone (no sweep beside it), outside the timed region: what an ideal overlap could hide
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        a.record()
        for _ in range(10):
            dist.all_reduce(sampler.rows)          # (the rows are all zero between sweeps: the sums stay zero)
        b.record()
        torch.cuda.synchronize()
        comm["allreduce_alone_ms"] = a.elapsed_time(b) / 10
    if dist is not None:
        t = torch.tensor([sites_local], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        total_sites = int(t.item())

    # N > 1: the same shard once more with the exchange pipelined over two document ranges (the rows of the first
    # range are all-reduced while the second is sampled) -- a probe, reported next to the timed line
    probe = None
    if dist is not None and name != "abstracts" and args.overlap < 0 and not args.no_extras:
        keep = (info["K"], info["V"], info["N"], info["live_topics"], info["docs_local"], info["docs_total"], info["desc"])
        del sampler, info
        torch.cuda.empty_cache()
        sampler, info = build_sampler(name, dev, rank, world, True, docs_total=args.docs,
                                      docs_per_group=args.docs_per_group, overlap=2)
        dt2, k2 = time_sweeps(sampler, max(5, args.steps // 4), 2, dist, dev)
        probe = {"overlap_ranges": 2, "steps": max(5, args.steps // 4), "ms_per_step": dt2 / max(5, args.steps // 4) * 1e3,
                 "exchange_ms": sampler.comm_stats(),
                 "sweeps": sampler.sweeps_done}
        psums = state_checksums(sampler)
        probe["state_checksum_n_k_after_probe"] = psums["n_k"]
        probe["checksum_matches_n1"], _ = checksum_verdict(name, info["docs_total"], sampler.sweeps_done, psums)

    trace("probes done")
    if rank == 0:
        K, V, N = info["K"], info["V"], info["N"]
        live = info["live_topics"]
        ms = dt / args.steps * 1e3
        value = total_sites * args.steps / dt / 1e6
        do_pmc = not args.no_pmc and name in PMC_WORKLOADS
        extras_on = world == 1 and not args.no_extras and name == "synth2" and not args.docs
        pmc_names = [w for w in PMC_WORKLOADS if w == name or extras_on] if do_pmc else []
        line = {
            "metric": "million tokens resampled/sec (Gibbs sweep)",
            "value": value, "unit": "Mtokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong" if name != "abstracts" else "weak",
            "vs_baseline": None, "dtype": "f64-exact (fp32 tier 0 + fp64 tiers: the drawn topic is the fp64 pipeline's)",
            "data": "synthetic" if name != "abstracts" else "tokenised abstracts_data.csv (fixture)",
            "config": {"workload": info["desc"], "docs_total": info["docs_total"], "docs_per_gpu": info["docs_local"],
                       "sites_per_doc": N, "K": K, "V": V, "alpha": ALPHA, "beta": BETA,
                       "label_mask": "dense" if live == K else "sparse (%.2f live topics per doc)" % live,
                       "kernel": "sparse" if sampler.live_off is not None else "dense",
                       "n_kw_rows": rows_description(sampler),
                       "n_kw_rows_short": rows_short(sampler),
                       "build_info": build_bits, "abi": _native.ABI_VERSION, "library": os.path.relpath(_native.LIB_PATH, ROOT),
                       "sites_per_sweep": total_sites, "tokens_per_site": 1.0 if name != "abstracts" else None, "timed_seconds": dt,
                       "exchange": sampler.exchange_description() if world > 1 else "none (single GPU: the commit log is "
                                   "folded straight into n_kw)",
                       "semantics": "per-document snapshot (bit-exact vs the reference under O3)",
                       "draw": "tiered: fp32 decision with a proven margin, fp64 / exact fp64 pipeline otherwise; "
                               "the result is the exact fp64 pipeline's",
                       "state_checksum_n_k": checksum, "state_checksum_n_kw": sums["n_kw"],
                       "sweeps_behind_checksum": sweeps_done},
            "draw_tiers": {"sites": int(sites_local) * (args.steps + args.warmup),
                           "fp32_tier_unsure": int(tier[1]), "exact_tier": int(tier[2])},
        }
        if world > 1:
            # the state after s sweeps is identical for every N: compare with the stored single-GPU digests
            line["checksum_matches_n1"], line["checksum_note"] = checksum_verdict(name, line["config"]["docs_total"],
                                                                                  sweeps_done, sums)
        if comm is not None:
            line["exchange_ms"] = comm
        if probe is not None:
            line["overlap_probe"] = probe
        # ---- extras (N = 1, default workload): the other configurations, timed in this run ----
        extra = {}
        measured = {name: dict(kernel_ms=kavg, sites=sites_local, docs=info["docs_local"], live=live,
                               shared=sampler._counts.numel() * 4)}
        if not args.no_cpu:
            # (N > 1: rank 0 times the port on the first documents of ITS shard while the other ranks wait at the final barrier)
            line["cpu_baseline"], line["speedup_vs_cpu_port"] = cpu_baseline_json(sampler, info, name, value)
        del sampler, info
        torch.cuda.empty_cache()
        if extras_on:
            for key, wname, st, wu in (("synth1", "synth1", 200, 5), ("dense_k256", "synth_k256", 100, 5), ("hbm_bound", "synth2_hostile", 40, 3),
                                       ("sparse_labels", "synth2_sparse", 100, 5),
                                       ("sparse_labels_colocated", "synth2_sparse_hier", 100, 5),
                                       ("sparse_labels_scrambled", "synth2_sparse_scr", 100, 5), ("abstracts", "abstracts", 3000, 20),
                                       ("wide_k2048", "synth_wide", 100, 5), ("wide_sparse_k2048", "synth_wide_sparse", 100, 5)):
                s2, i2 = build_sampler(wname, dev, 0, 1, False)
                torch.cuda.synchronize()
                dt2, k2 = time_sweeps(s2, st, wu, events=wname != "abstracts")
                v2 = s2.S * st / dt2 / 1e6
                e = {"workload": i2["desc"], "value": v2, "unit": "Mtokens/s", "steps": st, "warmup": wu,
                     "ms_per_step": dt2 / st * 1e3, "timed_seconds": dt2, "docs": i2["docs_local"], "sites_per_sweep": s2.S,
                     "K": i2["K"], "V": i2["V"], "kernel": "sparse" if s2.live_off is not None else "dense",
                     "n_kw_rows": rows_description(s2), "n_kw_rows_short": rows_short(s2), "kernel_ms": k2}
                measured[wname] = dict(kernel_ms=k2, sites=s2.S, docs=i2["docs_local"], live=i2["live_topics"],
                                       shared=s2._counts.numel() * 4)
                if wname == "abstracts":
                    # a site of a real corpus carries its word's frequency in the document (1.24 tokens on average here): both rates
                    tokens_per_site = float(s2.freq.sum(dtype=torch.int64).item()) / max(s2.S, 1)
                    e["unit"], e["Mtokens_s"], e["tokens_per_site"] = "Msites/s", v2 * tokens_per_site, tokens_per_site
                    e["live_topics_per_doc"] = i2["live_topics"]
                    e["algorithmic_GBps"] = algorithmic_bytes(s2.S, i2["docs_local"], i2["live_topics"]) / (k2 * 1e-3) / 1e9
                    e["note"] = "latency bound (one 48-site document chain per lane group), not bandwidth bound"
                    if not args.no_cpu:
                        e["cpu_baseline"], e["speedup_vs_cpu_port"] = cpu_baseline_json(s2, i2, wname, v2)
                        e["target_speedup"] = 50.0
                if wname == "synth_wide_sparse":
                    e["live_topics_per_doc"] = i2["live_topics"]
                    e["note"] = "the sparse-label kernel (one lane per allowed topic) on a layout with 16 pairwise leaves"
                if wname == "synth2_sparse_hier":
                    e["live_topics_per_doc"] = i2["live_topics"]
                    e["note"] = ("the same kernel and arithmetic as sparse_labels; only WHICH labels a document carries differs: "
                                 "siblings with neighbouring topic ids share cache lines (a caller gets this by ordering a "
                                 "hierarchical labelset by code before handing it to LabeledLDA)")
                if wname == "synth2_sparse_scr":
                    e["live_topics_per_doc"] = i2["live_topics"]
                    e["note"] = ("sparse_labels_colocated's label sets under a fixed permutation of the topic ids: the image's own column "
                                 "order (greedy clustering of the label co-occurrence) puts the siblings back into shared cache lines")
                if wname == "synth2_sparse":
                    e["live_topics_per_doc"] = i2["live_topics"]
                    e["algorithmic_GBps"] = algorithmic_bytes(s2.S, i2["docs_local"], i2["live_topics"]) / (k2 * 1e-3) / 1e9
                    e["note"] = ("one lane per allowed topic; every 4-byte gather of n_kw[v, pos] that misses the L2 costs a "
                                 "128-byte line fill (roofline.l2: fabric requests per site), random labels")
                if wname == "synth_wide":
                    e["kernel"] = "wide"
                    e["note"] = ("general path (one wavefront per document, fp32 tier 0 in front of the fp64 decision): bound by the "
                                 "latency of one wavefront's dependent chain at 8 wavefronts per CU (LDS: 20 KB each -- cached fp64 "
                                 "factors + int16 count changes)")
                extra[key] = e
                del s2, i2
                torch.cuda.empty_cache()
            extra["cascade"] = cascade_extra()
            extra["pipeline_abstracts"] = pipeline_extra(with_cpu=not args.no_cpu)
            extra["cascade_test"] = cascade_test_extra(with_cpu=not args.no_cpu)
        # ---- roofline: HBM-side counters collected in this run (separate rocprofv3 passes) ----
        pmc, source = ({}, "not collected (--no-pmc)")
        if pmc_names and world == 1:
            pmc, source = pmc_collect(pmc_names, keep_dir=args.pmc_keep or None, docs=args.docs)
        elif pmc_names:
            # N > 1: the counters of a single-process replica of rank 0's shard on rank 0's GPU (the other ranks idle at the final
            # barrier): the same kernel and n_kw, 1 / N of the documents -- three passes (fabric bytes and issue), not five
            pmc, source = pmc_collect(pmc_names, keep_dir=args.pmc_keep or None, docs=measured[name]["docs"],
                                      passes=[p_ for p_ in PMC_PASSES if p_[0] in ("fetch", "write", "sq")])
            source = "single-process replica of rank 0's shard (%d documents); " % measured[name]["docs"] + source
        m = measured[name]
        line["roofline"] = roofline_json(m["kernel_ms"], m["sites"], m["docs"], m["live"], pmc.get(name), source,
                                         stored_key=name, shared_bytes=m["shared"])
        for key, wname in (("synth1", "synth1"), ("dense_k256", "synth_k256"), ("hbm_bound", "synth2_hostile"), ("sparse_labels", "synth2_sparse"),
                           ("sparse_labels_colocated", "synth2_sparse_hier"), ("sparse_labels_scrambled", "synth2_sparse_scr"), ("wide_sparse_k2048", "synth_wide_sparse"), ("wide_k2048", "synth_wide"), ("abstracts", "abstracts")):
            if key in extra:
                m = measured[wname]
                extra[key]["roofline"] = roofline_json(m["kernel_ms"], m["sites"], m["docs"], m["live"], pmc.get(wname),
                                                       source, stored_key=wname, shared_bytes=m["shared"])
        if extra:
            line["extra"] = extra
        detail_path = None
        if args.detail_out:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.detail_out)), exist_ok=True)
                with open(args.detail_out, "w") as fh:
                    json.dump(line, fh, indent=1)
                detail_path = args.detail_out
            except OSError as e:                                  # the line must survive a read-only tree
                print("bench.py: could not write %s: %r" % (args.detail_out, e), file=sys.stderr)
        sys.stdout.flush()
        print(json.dumps(compact_line(line, detail_path)), flush=True)
    trace("line printed / waiting at the final barrier")
    if dist is not None:
        dist.barrier()
        trace("final barrier passed")
        dist.destroy_process_group()
    trace("done")
    watchdog.cancel()


if __name__ == "__main__":
    main()
