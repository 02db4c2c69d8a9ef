#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/c5
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -70 > gpurun_out/c5/pytest_gpu.log
tail -12 gpurun_out/c5/pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B > gpurun_out/c5/bench_ant4096_w2.json 2> gpurun_out/c5/bench.err
TDS_HIP_W2=0 $B > gpurun_out/c5/bench_ant4096_w1.json 2>> gpurun_out/c5/bench.err
$B --envs-per-gpu 2048 > gpurun_out/c5/bench_ant2048_w2.json 2>> gpurun_out/c5/bench.err
TDS_HIP_W2=0 $B --envs-per-gpu 2048 > gpurun_out/c5/bench_ant2048_w1.json 2>> gpurun_out/c5/bench.err
TDS_HIP_W2=2 $B --envs-per-gpu 8192 > gpurun_out/c5/bench_ant8192_w2forced.json 2>> gpurun_out/c5/bench.err
$B --envs-per-gpu 8192 > gpurun_out/c5/bench_ant8192_default.json 2>> gpurun_out/c5/bench.err
$B --model laikago_soft --envs-per-gpu 2048 > gpurun_out/c5/bench_laikago_soft2048_w2.json 2>> gpurun_out/c5/bench.err
TDS_HIP_W2=0 $B --model laikago_soft --envs-per-gpu 2048 > gpurun_out/c5/bench_laikago_soft2048_w1.json 2>> gpurun_out/c5/bench.err
$B --model pendulum5 --dtype f32 > gpurun_out/c5/bench_pendulum5_w2.json 2>> gpurun_out/c5/bench.err
TDS_HIP_W2=0 $B --model pendulum5 --dtype f32 > gpurun_out/c5/bench_pendulum5_w1.json 2>> gpurun_out/c5/bench.err
for f in gpurun_out/c5/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])" 2>&1 | tail -1)"; done
tail -3 gpurun_out/c5/bench.err
