#!/bin/bash
mkdir -p gpurun_out/r16
for x in 0 20000 60000 110000; do
echo "extra LDS $x" >> gpurun_out/r16/occ.txt
LLDA_EXTRA_LDS=$x python bench.py --steps 10 --warmup 2 --no-cpu --no-pmc --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['roofline']['kernel_ms'])" >> gpurun_out/r16/occ.txt
done
cat gpurun_out/r16/occ.txt
