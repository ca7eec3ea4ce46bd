#!/bin/bash
# How much does the 16-bit-row sweep kernel depend on its wave count?  Builds the library with dummy dynamic LDS on the THREE-wave
# form's launch (-DABL_EXTRA_LDS_BYTES=20000: two workgroups per CU instead of three) in the build container:
#   bash tools/occupancy_probe.sh build
# and times it against the in-tree library on the GPU box (debug: the four-wave form is the production one; the probe library is
# compared on workloads whose documents hold 2^16 tokens or more, or with LLDA_BENCH... see tools/ab_lib.py):
#   gpurun -- 'bash tools/occupancy_probe.sh run synth2 k1024'
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/bin
  (cd lda_thesis_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
      -I../../include -Wno-unused-function -DABL_EXTRA_LDS_BYTES=20000 -o ../../tools/bin/libllda_lds20000.so llda_gibbs.hip)
  ls -la tools/bin/libllda_lds20000.so
else
  shift
  for w in "$@"; do
    LLDA_GIBBS_LIB=$PWD/tools/bin/libllda_lds20000.so python tools/ab_lib.py $w 2>&1 | grep kernel
    python tools/ab_lib.py $w 2>&1 | grep kernel
  done
fi
