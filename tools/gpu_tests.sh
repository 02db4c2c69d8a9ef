#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/t
timeout 1500 python -m pytest tests -m gpu -q "$@" 2>&1 | tail -40 > gpurun_out/t/pytest_gpu.log
tail -40 gpurun_out/t/pytest_gpu.log | cut -c1-300
