set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r06a
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "quad_tier0" > gpurun_out/r06a/neartie_quad.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r06a/neartie_quad.log
timeout 1200 python tools/quad_margin_headroom.py > gpurun_out/r06a/headroom.md 2> gpurun_out/r06a/headroom.err; echo "rc=$?"; cat gpurun_out/r06a/headroom.md; tail -3 gpurun_out/r06a/headroom.err
