#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/c3
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/c3/pytest_gpu.log
tail -8 gpurun_out/c3/pytest_gpu.log
timeout 600 python tools/auto_reset_modes.py > gpurun_out/c3/auto_reset_modes_default.txt 2>&1
TDS_HIP_LIB=$PWD/tiny-differentiable-simulator_amd/libtds_hip_w2.so timeout 600 python tools/auto_reset_modes.py > gpurun_out/c3/auto_reset_modes_w2.txt 2>&1
timeout 600 python tools/rollout_modes.py > gpurun_out/c3/rollout_modes_default.txt 2>&1
TDS_HIP_LIB=$PWD/tiny-differentiable-simulator_amd/libtds_hip_w2.so timeout 600 python tools/rollout_modes.py > gpurun_out/c3/rollout_modes_w2.txt 2>&1
echo "--- auto reset default"; cat gpurun_out/c3/auto_reset_modes_default.txt
echo "--- auto reset w2"; cat gpurun_out/c3/auto_reset_modes_w2.txt
echo "--- rollout default"; cat gpurun_out/c3/rollout_modes_default.txt
echo "--- rollout w2"; cat gpurun_out/c3/rollout_modes_w2.txt
