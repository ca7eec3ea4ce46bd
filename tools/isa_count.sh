#!/bin/bash
# tools/isa_count.sh [mangled-name-substring]  --  device assembly of one kernel + instruction counts by unit
# (default: the K = 512 sweep kernel).  Output: /tmp/isa/kernel.s
set -e
SUB=${1:-llda_sweep_kernelILi32ELi16ELb0ELb1ELb1E}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$ROOT/include \
    -Wno-unused-function --cuda-device-only -S -o llda.s $ROOT/lda_thesis_amd/csrc/llda_gibbs.hip $EXTRA 2>/dev/null
start=$(grep -n "^_ZN.*${SUB}.*:" llda.s | head -1 | cut -d: -f1)
end=$(awk -v s=$start 'NR>s && /\.amdhsa_kernel|^\.Lfunc_end/ {print NR; exit}' llda.s)
sed -n "${start},${end}p" llda.s > kernel.s
echo "lines $start..$end"
echo "VALU  $(grep -cE '^\s+v_' kernel.s)   SALU $(grep -cE '^\s+s_' kernel.s)   DS $(grep -cE '^\s+ds_' kernel.s)   VMEM $(grep -cE '^\s+(global|scratch|buffer|flat)_' kernel.s)"
grep -E "vgpr_count|sgpr_count|spill|scratch" llda.s | grep -A0 -B0 . | head -0
awk -v s=$end 'NR>=s && NR<s+60' llda.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed_size" | head -3
