set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06c
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/r06c/dropin.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r06c/dropin.log
python - > gpurun_out/r06c/pipeline.txt 2>&1 <<'PY'
import json, sys, time, cProfile, pstats, io
sys.path.insert(0, ".")
import numpy as np, torch, bench
out = bench.pipeline_extra(with_cpu=True)
print(json.dumps({k: out[k] for k in ("value", "stages_s", "cold_first_pass_s", "speedup_vs_cpu_port")}, indent=1))
print(json.dumps(out["cpu_baseline"], indent=1))
# where the constructor's time goes
from lda_thesis_amd.LabeledLDA import LabeledLDA
from lda_thesis_amd.text import Dictionary
import os
g = np.load(os.path.join("tests", "golden", "abstracts_d3.npz"))
names = [str(x) for x in g["labelset"]]
docs, labs = bench._abstracts_tokens(g, "doc_off", "word", "freq", "lab_off", "lab_idx", names)
dicti = Dictionary(docs)
pr = cProfile.Profile()
np.random.seed(0)
pr.enable()
m = LabeledLDA(docs, labs, names[1:], dicti, 0.1, 0.01, seed=1)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue())
PY
grep -v amdgpu.ids gpurun_out/r06c/pipeline.txt | head -90
