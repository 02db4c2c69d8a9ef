#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/c8
for rep in 1 2; do
timeout 300 python tools/profile_phases.py ant 4096 0 100 > gpurun_out/c8/phases_ant4096_$rep.txt 2>&1
done
timeout 300 python tools/profile_phases.py ant 2048 0 100 > gpurun_out/c8/phases_ant2048.txt 2>&1
timeout 300 python tools/profile_phases.py pendulum5 4096 0 100 > gpurun_out/c8/phases_pendulum5.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c8/bench_ant4096.json 2> gpurun_out/c8/bench.err
cat gpurun_out/c8/phases_ant4096_1.txt
tail -30 gpurun_out/c8/phases_ant4096_2.txt
tail -3 gpurun_out/c8/bench_ant4096.json
