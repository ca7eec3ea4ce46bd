"""long-run check of the quad kernel (csrc/kernel_quad.hpp) against the kernel the sampler takes without it: the state digests after
many sweeps, with the library's per-sweep row flags and the hand-over policy live (quad=None vs quad=False).
python tools/quad_soak.py [sweeps [workload[:documents] ...]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ok = True
for spec in (sys.argv[2:] or ["synth1", "synth_k256:50000", "synth2:62500"]):
    name, _, docs = spec.partition(":")
    res = {}
    for quad in ("off", None):
        os.environ.pop("LLDA_QUAD", None)
        if quad:
            os.environ["LLDA_QUAD"] = quad
        s, info = bench.build_sampler(name, dev, 0, 1, False, docs_total=int(docs or 0))
        started = bool(s.quad)
        for _ in range(sweeps):
            s.sweep()
        s.check_status()
        torch.cuda.synchronize()
        wide = int((s.n_kw.max(dim=1).values > 65535).sum())
        res[quad] = (bench.state_checksums(s), s.z.clone(), started, bool(s.quad), wide, s.status.cpu().numpy()[:3].tolist())
        del s, info
        torch.cuda.empty_cache()
    same = res["off"][0] == res[None][0] and bool(torch.equal(res["off"][1], res[None][1]))
    ok = ok and same
    print("%s: %d sweeps | quad started %s, still on %s, wide rows at the end %d | status %s vs %s | same state: %s" %
          (spec, sweeps, res[None][2], res[None][3], res[None][4], res[None][5], res["off"][5], same), flush=True)
os.environ.pop("LLDA_QUAD", None)
print("ALL EQUAL" if ok else "MISMATCH")
