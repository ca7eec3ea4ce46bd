"""per-sweep time of llda_commit_log (the fold of the commit log into n_kw) next to the sweep kernel, HIP events around every native
call, without the exchange: python tools/fold_time.py [workload[:documents] ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from lda_thesis_amd import _native
dev = torch.device("cuda", 0)
names = ("pack_rows16_all", "sweep", "commit_log", "pack_image", "apply_delta")
log = []
orig = {n: getattr(_native, n) for n in names}
def wrap(n):
    def f(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig[n](*a, **k); e1.record(); log.append((n, e0, e1)); return r
    return f
for n in names:
    setattr(_native, n, wrap(n))
for spec in (sys.argv[1:] or ["synth2", "synth1", "synth2_sparse"]):
    name, _, docs = spec.partition(":")
    s, info = bench.build_sampler(name, dev, 0, 1, False, docs_total=int(docs or 0))
    for _ in range(3):
        s.sweep()
    torch.cuda.synchronize(); del log[:]
    for _ in range(10):
        s.sweep()
    torch.cuda.synchronize()
    parts = {}
    for n, e0, e1 in log:
        parts[n] = parts.get(n, 0.0) + e0.elapsed_time(e1) / 10
    print(spec, os.environ.get("LLDA_GIBBS_LIB", "production").split("/")[-1], {k: round(v, 4) for k, v in parts.items()}, bench.state_checksums(s), flush=True)
    del s, info; torch.cuda.empty_cache()
