#!/bin/bash
# round 3, run 9: fold-in with the decided tier: tests (drop-in suite), rates
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03i
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_readout.py -x -q > $OUT/dropin.log 2>&1; echo "dropin rc=$?" >> $OUT/dropin.log
tail -5 $OUT/dropin.log
python tools/bench_cascade.py --test-it 150 > $OUT/bench_cascade.json 2>/dev/null
python tools/bench_foldin.py --it 150 > $OUT/bench_foldin.json 2>/dev/null
cat $OUT/bench_foldin.json
python - <<'PY'
import json
b=json.load(open("gpurun_out/r03i/bench_cascade.json")); print(b["value"], b["test_down_tree"])
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03i_stats -o ct -- python $REPO/tools/prof_cascade_test.py > /tmp/r03i_ct.log 2>&1)
for f in $(find /tmp/r03i_stats -name "*_kernel_stats.csv"); do head -25 $f > $OUT/cascade_test_kernel_stats.csv; done
cat $OUT/cascade_test_kernel_stats.csv | cut -c1-150
