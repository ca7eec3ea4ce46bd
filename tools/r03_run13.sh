#!/bin/bash
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03p
mkdir -p $OUT
for k in 2048 1088 3000; do echo "== TOUCH" >> $OUT/abl.txt; LLDA_GIBBS_LIB=$REPO/tools/bin/libllda_abl_TOUCH.so python tools/abl_wide.py $k 2>&1 | grep "^K" >> $OUT/abl.txt; echo "== base" >> $OUT/abl.txt; python tools/abl_wide.py $k 2>&1 | grep "^K" >> $OUT/abl.txt; done
cat $OUT/abl.txt | cut -c1-200
