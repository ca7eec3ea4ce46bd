#!/bin/bash
# gpurun -- 'bash tools/n4_probe.sh [runs] [extra bench args]': the N = 4 code path of bench.py on a one-GPU box (four processes on cuda:0, a gloo
# group reducing CUDA tensors), several times over, with time-stamped progress and a stack dump of every rank that is still going after
# 45 s.  Logs: gpurun_out/n4r5/.  One summary line per attempt.
RUNS=${1:-6}; shift
OUT=gpurun_out/n4r5; mkdir -p $OUT
for i in $(seq 1 $RUNS); do
  t0=$(date +%s.%N)
  LLDA_BENCH_TRACE=1 LLDA_BENCH_TRACE_DUMP_S=45 timeout 150 python bench.py --gpus 4 --one-device --dist-backend gloo --steps 5 --warmup 2 --no-extras "$@" \
      > $OUT/line_$i.json 2> $OUT/err_$i.txt
  rc=$?
  t1=$(date +%s.%N); secs=$(python3 -c "print($t1 - $t0)")
  python3 - "$OUT/line_$i.json" "$rc" "$secs" <<'PY'
import json, sys
path, rc, secs = sys.argv[1], sys.argv[2], float(sys.argv[3])
try:
    d = json.loads(open(path).read().strip().split("\n")[-1])
    print("attempt %s: rc %s, %.1f s, value %.0f, ms_per_step %.1f, checksum_matches_n1 %s, error %s" %
          (path, rc, secs, d.get("value", -1), d.get("ms_per_step", -1), d.get("checksum_matches_n1"), d.get("error")))
except Exception as e:
    print("attempt %s: rc %s, %.1f s, NO LINE (%r)" % (path, rc, secs, e))
PY
done
