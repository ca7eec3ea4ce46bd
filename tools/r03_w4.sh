#!/bin/bash
mkdir -p gpurun_out/w4
export TMPDIR=/tmp
for e in 1 ""; do
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
tag=$(echo $set | cut -d' ' -f1)
(cd /tmp && LLDA_NO_W4=$e rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/prof_${tag}_$e -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-pmc --no-extras > /dev/null 2>&1)
f=$(find /tmp/prof_${tag}_$e -name "*counter_collection.csv" | head -1)
python - "$f" "NO_W4=$e" <<'PY' >> gpurun_out/w4/pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, set()])
for r in csv.DictReader(open(sys.argv[1])):
    if "llda_sweep_kernel" in r["Kernel_Name"]:
        k = r["Counter_Name"]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1].add(r["Dispatch_Id"])
for k in sorted(acc):
    print(sys.argv[2], k, "%.6g per launch (%d launches)" % (acc[k][0] / len(acc[k][1]), len(acc[k][1])))
PY
done
done
cat gpurun_out/w4/pmc.txt
