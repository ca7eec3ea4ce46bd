set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06b
{
bash tools/ab_lib.sh synth2 synth1 synth_k256
} > gpurun_out/r06b/ab_hot.txt 2>&1
cat gpurun_out/r06b/ab_hot.txt | grep -v amdgpu.ids
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad or headline or narrow" > gpurun_out/r06b/quad_tests.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r06b/quad_tests.log
