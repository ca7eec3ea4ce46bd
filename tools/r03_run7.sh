#!/bin/bash
# round 3, run 7: where does a site of the fp32-tiered wide kernel spend its time?  (ablation libraries: wrong results, timing only)
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03g
mkdir -p $OUT
for v in nomargin NOADDLOAD NOROW; do
  for k in 2048 1088; do echo "== $v" >> $OUT/abl.txt; LLDA_GIBBS_LIB=$REPO/tools/bin/libllda_abl_$v.so python tools/abl_wide.py $k 2>&1 | grep "^K" >> $OUT/abl.txt; done
done
cat $OUT/abl.txt
timeout 600 bash tools/pmc_probe.sh synth_wide > $OUT/pmc_synth_wide.txt 2>&1
cat $OUT/pmc_synth_wide.txt | head -50
