#!/bin/bash
# How much does the THREE-wave 16-bit-row sweep kernel depend on its wave count?  (Measured before the four-wave form existed: 67.0 ms
# at two waves per SIMD against 50.9 at three on configs[3] -- the measurement that made the case for the fourth wave)
# Builds the library with dummy dynamic LDS on that kernel's launch (-DABL_EXTRA_LDS_BYTES=20000: two workgroups per CU) in the build
# container:            bash tools/occupancy_probe.sh build
# and times it against the in-tree library on the GPU box:    gpurun -- 'bash tools/occupancy_probe.sh run <workload> ...'
# The dummy LDS only reaches the three-wave form, which runs where a document holds 2^16 tokens or more (else the production library
# runs the four-wave form and the pair compares two waves with four).
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
