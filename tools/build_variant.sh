#!/bin/bash
# tools/build_variant.sh <name> [-D...]  -- an ablation / experiment build of the library into tools/bin/libllda_<name>.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p $ROOT/tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -I$ROOT/include \
    -Wno-unused-function "$@" -o $ROOT/tools/bin/libllda_$name.so $ROOT/lda_thesis_amd/csrc/llda_gibbs.hip
