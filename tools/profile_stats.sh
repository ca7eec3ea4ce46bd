#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench's timed line (and of the sparse-label workloads) on the GPU box:
#     gpurun --timeout 1500 -- 'bash tools/profile_stats.sh <tag>'     -> gpurun_out/prof_<tag>/*_kernel_stats.csv
# Every rocprofv3 call carries --output-format csv (the default rocpd output never finished on this pool) and its own timeout.
# The sweep kernel's average in default_kernel_stats.csv must agree with roofline.kernel_ms of the bench line (HIP events).
set -u
TAG=${1:?tag}
export TMPDIR=/tmp
REPO=$(pwd)
P=/tmp/prof_$TAG
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $P
run() {   # name, bench arguments...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/$name -o $name -- \
      python $REPO/bench.py "$@" --no-cpu --no-pmc --no-extras --detail-out "" > $P/$name.log 2>&1)
  echo "$name rc=$?"
  for f in $(find $P/$name -name "*_kernel_stats.csv"); do cp $f $OUT/${name}_kernel_stats.csv; done
  tail -n 1 $P/$name.log | head -c 600; echo
  head -6 $OUT/${name}_kernel_stats.csv
}
run default --gpus 1 --steps 20 --warmup 5
run sparse --workload synth2_sparse --steps 50 --warmup 5
run scrambled --workload synth2_sparse_scr --steps 50 --warmup 5
run abstracts --workload abstracts --steps 500 --warmup 20
run synth1 --workload synth1 --steps 200 --warmup 5
# the same lines WITH the counter passes, raw per-dispatch counter rows kept
timeout 600 python bench.py --workload synth2_sparse --steps 50 --warmup 5 --no-cpu --pmc-keep $OUT/pmc_sparse --detail-out $OUT/sparse_detail.json > $OUT/sparse_line.json 2> $OUT/sparse.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --pmc-keep $OUT/pmc_default --detail-out $OUT/default_detail.json > $OUT/default_line.json 2> $OUT/default.err
du -sh $OUT
