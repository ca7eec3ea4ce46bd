set -u
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q > gpurun_out/r03a/multirank.log 2>&1; echo "multirank rc=$?" >> gpurun_out/r03a/multirank.log
tail -5 gpurun_out/r03a/multirank.log
rocprofv3 -L > gpurun_out/r03a/counters_list.txt 2>&1
for w in synth2_sparse synth_wide_sparse synth_wide; do
  timeout 600 bash tools/pmc_probe.sh $w > gpurun_out/r03a/pmc_$w.txt 2>&1
done
timeout 600 bash tools/pmc_probe.sh synth2 250000 > gpurun_out/r03a/pmc_synth2_250k.txt 2>&1
timeout 600 bash tools/pmc_probe.sh synth2_hostile > gpurun_out/r03a/pmc_synth2_hostile.txt 2>&1
tail -30 gpurun_out/r03a/pmc_synth2_sparse.txt
