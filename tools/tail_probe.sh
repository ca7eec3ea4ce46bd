#!/bin/bash
# kernel time of the dense sweep vs the number of equal-length documents (rounds of resident lane groups): synth1
for D in 24576 49152 98304 100000 110592 122880; do
  python bench.py --workload synth1 --docs $D --steps 50 --warmup 5 --no-cpu --no-pmc --no-extras 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print($D, 'ms/step %.4f' % j['ms_per_step'], 'kernel %.4f' % j['roofline']['kernel_ms'], 'Msites/s %.0f' % j['value'])"
done
