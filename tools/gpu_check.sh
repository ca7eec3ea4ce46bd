#!/bin/bash
# EXACTLY what the driver runs at the end of a round, on the GPU box, from the repo root:
#     gpurun --timeout 3000 -- 'bash tools/gpu_check.sh'
# smoke(), `pytest -m gpu -x -q` (no path argument: pytest.ini's testpaths decides what is collected), the bench line with the
# driver's arguments.  Logs under gpurun_out/check/.  The last line of bench.out must parse and stay under 8 KB.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/check
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 2700 python -m pytest -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gputest rc=$?"; grep -E "passed|failed|error" $OUT/gputest.log | tail -3
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc=$?"
cp gpurun_out/bench_detail.json $OUT/bench_detail.json 2>/dev/null
tail -n 1 $OUT/bench.out > $OUT/bench_line.json
python3 - <<'PY'
import json
t = open("gpurun_out/check/bench_line.json").read()
d = json.loads(t)
print("bench line: %d bytes, value %.1f %s, ms_per_step %.2f, roofline.frac %s, cpu_baseline %s, extras %d" % (
    len(t), d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], len(d.get("extra", {}))))
assert len(t) < 8000
PY
