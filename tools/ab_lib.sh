#!/bin/bash
# gpurun -- 'bash tools/ab_lib.sh'  : tools/bin/libllda_base.so (built from the previous commit) against the in-tree library
for w in "$@"; do
  LLDA_GIBBS_LIB=$PWD/tools/bin/libllda_base.so python tools/ab_lib.py $w
  python tools/ab_lib.py $w
done
