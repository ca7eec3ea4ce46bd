#!/bin/bash
# What the driver runs at the end of a round, on the GPU box:  gpurun -- 'bash tools/gpu_check.sh'
# smoke(), the -m gpu tests, the default bench line.  Logs under gpurun_out/check/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/check
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gputest rc=$?"; grep -E "passed|failed" $OUT/gputest.log | tail -1
timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; head -c 400 $OUT/bench_default.json; echo
