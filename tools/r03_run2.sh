#!/bin/bash
# round 3, run 2: the default bench line with the new counter passes and (f)-row extras, kernel stats of the same command,
# the gather micro-benchmark (with FETCH_SIZE), and the N = 1 state digests for the N > 1 self-check.
set -u
export TMPDIR=/tmp
REPO=$(pwd)
OUT=$REPO/gpurun_out/${TAG:-r03b}
mkdir -p $OUT
timeout 1500 python bench.py --pmc-keep $OUT/pmc > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -c 600 $OUT/bench_default.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${TAG:-r03b}_stats -o default -- \
    python $REPO/bench.py --steps 10 --warmup 2 --no-cpu --no-pmc > /tmp/${TAG:-r03b}_stats.log 2>&1)
for f in $(find /tmp/${TAG:-r03b}_stats -name "*_kernel_stats.csv"); do cp $f $OUT/stats_kernel_stats.csv; done
tools/bin/gather_ubench > $OUT/gather_ubench.txt 2>&1
for cfg in "8192 1" "205 1" "8192 16"; do
  set -- $cfg
  (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/r03b_g_$1_$2 -o g -- $REPO/tools/bin/gather_ubench one $1 $2 > $OUT/gather_one_$1_$2.json 2>/tmp/g.err)
  python - $1 $2 $OUT <<'PY'
import csv, glob, sys
fp, w, out = sys.argv[1], sys.argv[2], sys.argv[3]
vals = []
for f in glob.glob("/tmp/r03b_g_%s_%s/**/*counter_collection.csv" % (fp, w), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            vals.append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
open("%s/gather_one_%s_%s.fetch.txt" % (out, fp, w), "w").write("\n".join("FETCH_SIZE_KB %.0f kernel_ns %d" % v for v in vals) + "\n")
PY
done
timeout 600 python bench.py --make-checksums 64 > $OUT/checksums_synth2.json 2> $OUT/checksums.err
du -sh $OUT; ls -la $OUT $OUT/pmc
