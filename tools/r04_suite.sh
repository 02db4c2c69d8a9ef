#!/bin/bash
# Runs ON THE GPU BOX: the whole GPU suite (xdist workers, per-test limit).
export TMPDIR=/tmp
O=gpurun_out/r04n
P=gpurun_out/profiles
mkdir -p $O $P
timeout 500 python -m pytest tests -m gpu -q -n 4 --timeout 120 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_gpu.log | cut -c1-300
cp $O/pytest_gpu.log $P/${OUT:-r04f_pytest_gpu}.log
