#!/bin/bash
# round 3, run 5: what the fp32 tier of the wide kernel costs without its fall-backs (ablation library: margin ~0, results
# are NOT the reference's), fold-in with pipelined memory operations: tests + rates
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03e
mkdir -p $OUT
for k in 2048 1088 3000 4296; do LLDA_GIBBS_LIB=$REPO/tools/bin/libllda_abl_nomargin.so python tools/abl_wide.py $k >> $OUT/abl_wide_nomargin.txt 2>&1; done
grep "^K" $OUT/abl_wide_nomargin.txt
timeout 1800 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_parity.py -x -q > $OUT/gputest.log 2>&1; echo "gputest rc=$?" >> $OUT/gputest.log
tail -4 $OUT/gputest.log
python tools/bench_cascade.py --test-it 150 > $OUT/bench_cascade.json 2>/dev/null
python tools/bench_foldin.py --it 150 > $OUT/bench_foldin.json 2>/dev/null
cat $OUT/bench_foldin.json
