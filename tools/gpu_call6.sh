#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/c6
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/c6/pytest_gpu.log
tail -6 gpurun_out/c6/pytest_gpu.log
timeout 2400 bash tools/collect_profiles.sh r02b > gpurun_out/c6/collect.log 2>&1
tail -80 gpurun_out/c6/collect.log
