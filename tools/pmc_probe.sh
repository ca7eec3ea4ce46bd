#!/bin/bash
# Where do the wave cycles of the dense sweep kernel go?  SQ / TA / TCC counters for one workload (separate passes).
#   tools/pmc_probe.sh <workload> [docs]     (through gpurun; prints per-launch means of the sweep kernel)
W=${1:?workload}; DOCS=${2:-0}
export TMPDIR=/tmp
REPO=$(pwd)
P=/tmp/pmcprobe_$W
rm -rf $P; mkdir -p $P
B="python $REPO/bench.py --workload $W --docs $DOCS --steps 3 --warmup 1 --no-cpu --no-pmc --no-extras"
pass() { n=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/$n -o p -- $B > $P/$n.log 2>&1) || echo "pass $n failed"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
pass b SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VALU SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass d TCC_HIT_sum TCC_MISS_sum
pass e TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum
pass f GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
pass g TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum
pass h TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum
pass i FETCH_SIZE
pass j WRITE_SIZE TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$P/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "llda_sweep" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg["kernel_ns (under counters)"].append(float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k in sorted(agg):
    v = agg[k]
    print("%-34s %.6g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
grep -l "failed\|rror" $P/*.log 2>/dev/null | head
