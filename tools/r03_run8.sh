#!/bin/bash
# round 3, run 8: wide fp32-tier kernel after the lane-0 reordering / single row buffer beyond two tiers: tests + rates, then the
# whole GPU suite
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/${TAG:-r03h}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or near_ties or seeded_inputs or sweeps_match" > $OUT/widetest.log 2>&1; echo "widetest rc=$?" >> $OUT/widetest.log
tail -3 $OUT/widetest.log
for k in 2048 1088 3000 4296 7688; do python tools/abl_wide.py $k 2>&1 | grep "^K" >> $OUT/abl_wide.txt; done
cat $OUT/abl_wide.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gputest rc=$?" >> $OUT/gputest.log
tail -4 $OUT/gputest.log
python tools/bench_cascade.py --test-it 150 > $OUT/bench_cascade.json 2>/dev/null
python tools/bench_foldin.py --it 150 > $OUT/bench_foldin.json 2>/dev/null
cat $OUT/bench_foldin.json
