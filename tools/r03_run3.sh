#!/bin/bash
# round 3, run 3: full GPU test suite with the re-registered sparse kernels, N = 2 self-check of the bench line,
# gather micro-benchmark (pair mode), sparse workloads
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/r03c
mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gputest rc=$?" >> $OUT/gputest.log
tail -4 $OUT/gputest.log
timeout 900 python bench.py --gpus 2 --dist-backend gloo --one-device --steps 5 --warmup 2 --no-extras > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
tail -c 300 $OUT/bench_2rank_gloo.err
tools/bin/gather_ubench > $OUT/gather_ubench.txt 2>&1
for cfg in "8192 2" "8192 1"; do
  set -- $cfg
  (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/r03c_g_$1_$2 -o g -- $REPO/tools/bin/gather_ubench one $1 $2 > $OUT/gather_one_$1_$2.json 2>/tmp/g.err)
  python - $1 $2 $OUT <<'PY'
import csv, glob, sys
fp, w, out = sys.argv[1], sys.argv[2], sys.argv[3]
vals = []
for f in glob.glob("/tmp/r03c_g_%s_%s/**/*counter_collection.csv" % (fp, w), recursive=True):
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Dispatch_Id"]
            agg[k] = (agg.get(k, (0, 0))[0] + float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    vals = list(agg.values())
open("%s/gather_one_%s_%s.fetch.txt" % (out, fp, w), "w").write("\n".join("FETCH_SIZE_KB %.0f kernel_ns %d" % v for v in vals) + "\n")
PY
done
python bench.py --workload synth2_sparse --no-pmc --no-cpu > $OUT/bench_synth2_sparse.json 2>/dev/null
python bench.py --workload synth_wide_sparse --no-pmc --no-cpu > $OUT/bench_synth_wide_sparse.json 2>/dev/null
python bench.py --workload abstracts --no-pmc --no-cpu --steps 3000 --warmup 20 > $OUT/bench_abstracts.json 2>/dev/null
python tools/bench_cascade.py > $OUT/bench_cascade.json 2>/dev/null
du -sh $OUT
