#!/bin/bash
# Collect the judged artifacts for one kernel version on the GPU box:
#   tools/profile.sh <tag>      (run through gpurun; writes gpurun_out/prof_<tag>/)
# then, back in the container:  python profiles/summarize.py <tag> synth2 125000 ; ... synth1 100000
# Counters are collected in their own passes (kernel trace only), one --pmc group per pass.
set -u
TAG=${1:?tag}
export TMPDIR=/tmp
P=/tmp/prof_$TAG
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> /dev/null
for W in synth2 synth1; do
    B="python bench.py --workload $W --steps 5 --warmup 1 --no-cpu"
    rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_$W -o $W -- $B > $P.log 2>&1
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch_$W -o $W -- $B >> $P.log 2>&1
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write_$W -o $W -- $B >> $P.log 2>&1
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
        SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $P/sq_$W -o $W -- $B >> $P.log 2>&1
    for k in stats fetch write sq; do
        for f in $(find $P/${k}_$W -name "*.csv" | grep -v agent_info | grep -v kernel_trace); do
            cp $f $OUT/${k}_$(basename $f)
        done
    done
done
python bench.py --workload synth1 > $OUT/bench_synth1.json 2> /dev/null
python bench.py --workload abstracts > $OUT/bench_abstracts.json 2> /dev/null
python bench.py --workload synth2_sparse > $OUT/bench_synth2_sparse.json 2> /dev/null
ls $OUT | wc -l
