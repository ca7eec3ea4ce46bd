"""per-sweep wall times of bench.py's extra "wide_k2048" (synth_wide, dense K = 2 048) in a fresh process: where does the extra
millisecond per step of some runs come from?  python tools/wide_step_probe.py [steps]"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
s, info = bench.build_sampler("synth_wide", dev, 0, 1, False)
torch.cuda.synchronize()
# the bench's measurement: 2 warm-up sweeps, 20 timed sweeps between two synchronisations
dt, k = bench.time_sweeps(s, 20, 2)
print("as the bench: %.3f ms per step, kernel %.3f ms" % (dt / 20 * 1e3, k))
# every sweep on its own
per = []
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.sweep()
    torch.cuda.synchronize()
    per.append((time.perf_counter() - t0) * 1e3)
print("per sweep (sync after each): " + " ".join("%.2f" % x for x in per))
# enqueue-only cost of a sweep on the host
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    s.sweep()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("20 sweeps: host enqueue %.3f ms per sweep, drained after %.3f ms per sweep; gc counts %s" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3, gc.get_count()))
# the bench process's situation: a heap full of small objects (the abstracts token lists and the numpy port's state), collector ON
if "--heap" in sys.argv:
    import bench as B
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "abstracts_d3.npz"))
    names = [str(x) for x in g["labelset"]]
    keep = [B._abstracts_tokens(g, "doc_off", "word", "freq", "lab_off", "lab_idx", names) for _ in range(12)]   # ~ 10 M objects
    events = []
    t_start = [0.0]
    def cb(phase, info):
        if phase == "start":
            t_start[0] = time.perf_counter()
        else:
            events.append((info["generation"], (time.perf_counter() - t_start[0]) * 1e3))
    gc.callbacks.append(cb)
    for rep in range(3):
        junk = [[i] for i in range(200000)]              # allocations between measurements, as the bench's own bookkeeping makes
        del events[:]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            s.sweep()
            junk.append([object() for _ in range(3000)])  # (a sweep itself allocates a few hundred objects: args structs, tensors views)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        print("collector on, big heap: %.3f ms per step; collections inside the timed region: %s" %
              (dt, ", ".join("gen%d %.1f ms" % e for e in events) or "none"))
    gc.callbacks.remove(cb)
    dt, k = bench.time_sweeps(s, 20, 2)
    print("bench.time_sweeps on the same heap (collector off inside): %.3f ms per step" % (dt / 20 * 1e3))
