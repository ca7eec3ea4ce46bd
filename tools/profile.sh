#!/bin/bash
# Collect the judged artifacts for one library version on the GPU box:
#   gpurun -- 'bash tools/profile.sh <tag>'      (writes gpurun_out/prof_<tag>/)
# then, back in the container:  python profiles/summarize.py <tag>
# bench.py collects the PMC counters itself (separate rocprofv3 --pmc passes, kernel trace only: fetch | write | sq | l2 | ta) and
# keeps one row per sweep-kernel dispatch and counter with --pmc-keep.  The --stats passes time the same command without
# counters: the timed line alone, then with the extras.  The gather micro-benchmark (tools/gather_ubench.hip -> tools/bin/gather_ubench, built in the container)
# calibrates FETCH_SIZE for 4-byte gathers.
set -u
TAG=${1:?tag}
export TMPDIR=/tmp
REPO=$(pwd)
P=/tmp/prof_$TAG
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT $P
timeout 1500 python bench.py --pmc-keep $OUT/pmc > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -c 400 $OUT/bench_default.err
# (i) the timed line alone: the dominant kernel's average here must agree with roofline.kernel_ms of the bench line
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o default -- \
    python $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-pmc --no-extras > $P/stats.log 2>&1)
for f in $(find $P/stats -name "*_kernel_stats.csv"); do cp $f $OUT/stats_kernel_stats.csv; done
# (ii) the same command with the extras: every other kernel of the library (the hot kernel's average then mixes workloads)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_x -o extras -- \
    python $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-pmc > $P/stats_x.log 2>&1)
for f in $(find $P/stats_x -name "*_kernel_stats.csv"); do cp $f $OUT/stats_extras_kernel_stats.csv; done
if [ -x tools/bin/gather_ubench ]; then
  tools/bin/gather_ubench > $OUT/gather_ubench.txt 2>&1
  for cfg in "8192 1" "8192 2" "205 1"; do
    set -- $cfg
    (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/g_$1_$2 -o g -- $REPO/tools/bin/gather_ubench one $1 $2 > $OUT/gather_one_$1_$2.json 2>/dev/null)
    python - $1 $2 $OUT $P <<'PY'
import csv, glob, sys
fp, w, out, p = sys.argv[1:5]
agg = {}
for f in glob.glob("%s/g_%s_%s/**/*counter_collection.csv" % (p, fp, w), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Dispatch_Id"]
            agg[k] = (agg.get(k, (0, 0))[0] + float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
open("%s/gather_one_%s_%s.fetch.txt" % (out, fp, w), "w").write("\n".join("FETCH_SIZE_KB %.0f kernel_ns %d" % v for v in agg.values()) + "\n")
PY
  done
fi
for k in 1088 2048 3000 4296 7688; do python tools/abl_wide.py $k 2>&1 | grep "^K" >> $OUT/abl_wide.txt; done
python tools/bench_cascade.py --test-it 150 > $OUT/bench_cascade.json 2> /dev/null
python tools/bench_foldin.py --it 150 > $OUT/bench_foldin.json 2> /dev/null
timeout 900 python bench.py --gpus 2 --dist-backend gloo --one-device --steps 5 --warmup 2 --no-extras > $OUT/bench_2rank_gloo_one_device.json 2> /dev/null
timeout 600 python bench.py --make-checksums 64 > $OUT/checksums_synth2.json 2> /dev/null
# the timed line with int32 rows (the 16-bit image off): the fabric-bound form of the kernel, with its counters
mkdir -p $OUT/int32
LLDA_BENCH_ROWS16=off timeout 900 python bench.py --no-cpu --no-extras --pmc-keep $OUT/int32/pmc > $OUT/int32/bench.json 2> /dev/null
# int32 rows against 16-bit rows on this library, states compared (tools/abl_rows16.py)
timeout 900 python tools/abl_rows16.py synth2 k1024 synth2_hostile zipf_v20k zipf_v300k 2>&1 | grep "|" > $OUT/abl_rows16.txt
du -sh $OUT; ls $OUT $OUT/pmc
