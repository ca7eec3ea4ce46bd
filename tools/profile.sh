#!/bin/bash
# Collect the judged artifacts for one kernel version on the GPU box:
#   tools/profile.sh <tag>      (run through gpurun; writes gpurun_out/prof_<tag>/)
# then, back in the container:  python profiles/summarize.py <tag>
# bench.py collects the PMC counters itself (separate rocprofv3 --pmc passes, kernel trace only); the raw counter CSVs
# are kept with --pmc-keep.  The --stats pass below times the same command without counters.
set -u
TAG=${1:?tag}
export TMPDIR=/tmp
REPO=$(pwd)
P=/tmp/prof_$TAG
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $P
python bench.py --pmc-keep $OUT/pmc > $OUT/bench_default.json 2> $OUT/bench_default.err
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o default -- \
    python $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-pmc --no-extras > $P/stats.log 2>&1)
for f in $(find $P/stats -name "*_kernel_stats.csv"); do cp $f $OUT/stats_kernel_stats.csv; done
python bench.py --workload synth2_sparse --no-pmc > $OUT/bench_synth2_sparse.json 2> /dev/null
python tools/bench_cascade.py > $OUT/bench_cascade.json 2> /dev/null
python tools/bench_cascade.py --one-by-one --reps 2 > $OUT/bench_cascade_one_by_one.json 2> /dev/null
ls $OUT $OUT/pmc
