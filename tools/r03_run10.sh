#!/bin/bash
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03j
mkdir -p $OUT
python tools/prof_cascade_test.py > $OUT/prof_cascade_test.txt 2>&1
head -70 $OUT/prof_cascade_test.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r03j_stats -o ct -- python tools/prof_cascade_test.py > /tmp/r03j_ct.log 2>&1
for f in $(find /tmp/r03j_stats -name "*_kernel_stats.csv"); do head -20 $f | cut -c1-160 > $OUT/cascade_test_kernel_stats.csv; done
cat $OUT/cascade_test_kernel_stats.csv
