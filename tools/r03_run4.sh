#!/bin/bash
# round 3, run 4: the fp32-tiered wide kernel (ABI 14) -- wide tests first, then the whole GPU suite, then rates
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03d
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or near_ties or seeded_inputs or sweeps_match" > $OUT/widetest.log 2>&1; echo "widetest rc=$?" >> $OUT/widetest.log
tail -15 $OUT/widetest.log
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gputest rc=$?" >> $OUT/gputest.log
tail -4 $OUT/gputest.log
python bench.py --workload synth_wide --no-pmc --no-cpu --steps 20 --warmup 2 > $OUT/bench_synth_wide.json 2>$OUT/bench_synth_wide.err
for k in 2048 1088 3000 4296 7688; do python tools/abl_wide.py $k >> $OUT/abl_wide.txt 2>&1; done
python tools/bench_cascade.py --test-it 150 > $OUT/bench_cascade.json 2>/dev/null
tail -5 $OUT/abl_wide.txt
