#!/bin/bash
# round 3, run 6: fp32 tier 0 of the wide kernel with fp64 factors in LDS -- tests, rates, and the rate without fall-backs
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03f
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or near_ties or seeded_inputs or sweeps_match" > $OUT/widetest.log 2>&1; echo "widetest rc=$?" >> $OUT/widetest.log
tail -5 $OUT/widetest.log
for k in 2048 1088 3000 4296 7688; do python tools/abl_wide.py $k >> $OUT/abl_wide.txt 2>&1; done
for k in 2048 1088 3000 4296; do LLDA_GIBBS_LIB=$REPO/tools/bin/libllda_abl_nomargin.so python tools/abl_wide.py $k >> $OUT/abl_wide_nomargin.txt 2>&1; done
grep "^K" $OUT/abl_wide.txt; echo; grep "^K" $OUT/abl_wide_nomargin.txt
