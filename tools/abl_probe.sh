#!/bin/bash
# ablations of the dense sweep kernel (tools/abl/*.so built with make OUT=... EXTRA=-DABL_x): what does the time pay for?
for W in "synth1 0" "synth2 125000"; do
  set -- $W
  for L in "" NOLOAD; do
    LLDA_GIBBS_LIB=${L:+$(pwd)/tools/abl/libabl_$L.so} python tools/abl_time.py $1 $2 2>&1 | tail -1
  done
done
