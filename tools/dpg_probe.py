"""documents a lane group walks per workgroup (llda_sweep_args.docs_per_group): kernel time of the sweep for a few values.
python tools/dpg_probe.py [workload[:documents] ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
for spec in (sys.argv[1:] or ["synth2:250000", "synth1"]):
    name, _, docs = spec.partition(":")
    for dpg in (1, 2, 3, 4, 8):
        s, info = bench.build_sampler(name, dev, 0, 1, False, docs_total=int(docs or 0), docs_per_group=dpg)
        dt, kms = bench.time_sweeps(s, 20, 3)
        print("%-16s docs_per_group %d  kernel %.4f ms  step %.4f ms  %s" % (spec, dpg, kms, dt / 20 * 1e3, bench.state_checksums(s)), flush=True)
        del s, info
        torch.cuda.empty_cache()
