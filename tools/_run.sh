set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06e
{
for w in synth1 "synth1 393216" synth_k100; do
  python tools/ab_lib.py $w
  LLDA_GIBBS_LIB=$PWD/tools/bin/libllda_k128.so python tools/ab_lib.py $w
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06e/ab_k128.txt
cat gpurun_out/r06e/ab_k128.txt
