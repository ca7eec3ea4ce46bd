"""The per-sweep parts of the multi-GPU path on ONE GPU, for the projection of DESIGN.md section 7: a shard of 1 M / N documents of
configs[3] with the exchange rows forced on (exchange_always: the commit log is folded into the packed rows, the rows are decoded into
the counts; the all-reduce itself is skipped -- there is no peer), every native call bracketed by HIP events.
python tools/exchange_parts.py [documents ...]   ->  one JSON line per shard size (ms per sweep, mean of 10)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    from lda_thesis_amd import _native, sampler as S
    dev = torch.device("cuda", 0)
    names = ("pack_rows16_all", "pack_rows16", "sweep", "commit_log", "apply_rows", "apply_delta")
    log = []
    orig = {n: getattr(_native, n) for n in names}

    def wrap(n):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig[n](*a, **k)
            e1.record()
            log.append((n, e0, e1))
            return r
        return f
    for n in names:
        setattr(_native, n, wrap(n))
    for docs in [int(x) for x in sys.argv[1:]] or [1000000, 500000, 250000, 125000]:
        s, info = bench.build_sampler("synth2", dev, 0, 1, False, docs_total=docs, force_exchange=True)
        assert s.rows is not None and s.quad
        for _ in range(3):
            s.sweep()
        torch.cuda.synchronize()
        del log[:]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            s.sweep()
        b.record()
        torch.cuda.synchronize()
        parts = {}
        for n, e0, e1 in log:
            parts[n] = parts.get(n, 0.0) + e0.elapsed_time(e1) / 10
        out = {"docs": docs, "sites": s.S, "ms_per_sweep_total": a.elapsed_time(b) / 10, "parts_ms": {k: round(v, 4) for k, v in parts.items()},
               "exchange_bytes": int(s.rows.numel() * 4), "rows": s.exchange_description()}
        print(json.dumps(out), flush=True)
        del s, info
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
