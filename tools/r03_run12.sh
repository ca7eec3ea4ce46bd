#!/bin/bash
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03l
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/smoke.log; tail -2 $OUT/smoke.log
# does RCCL accept two ranks on one device?
timeout 300 python bench.py --gpus 2 --one-device --docs 200000 --steps 3 --warmup 1 --no-extras > $OUT/bench_2rank_nccl_one_device.json 2> $OUT/bench_2rank_nccl_one_device.err; echo "nccl one-device rc=$?"
tail -c 1500 $OUT/bench_2rank_nccl_one_device.err
head -c 600 $OUT/bench_2rank_nccl_one_device.json
