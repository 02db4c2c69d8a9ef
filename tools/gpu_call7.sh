#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/c7
B="timeout 300 python bench.py --no-cpu-baseline"
for rep in 1 2; do
$B > gpurun_out/c7/bench_ant4096_w2_$rep.json 2> gpurun_out/c7/bench.err
TDS_HIP_GRAM=1 $B > gpurun_out/c7/bench_ant4096_w2gram_$rep.json 2>> gpurun_out/c7/bench.err
TDS_HIP_W2=0 $B > gpurun_out/c7/bench_ant4096_w1_$rep.json 2>> gpurun_out/c7/bench.err
done
$B --envs-per-gpu 2048 > gpurun_out/c7/bench_ant2048_w2.json 2>> gpurun_out/c7/bench.err
TDS_HIP_GRAM=1 $B --envs-per-gpu 2048 > gpurun_out/c7/bench_ant2048_w2gram.json 2>> gpurun_out/c7/bench.err
TDS_HIP_GRAM=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_f32.py -m gpu -q -k "golden or closed_loop or full_size or overflow or mixed" 2>&1 | tail -8
for f in gpurun_out/c7/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'])" 2>&1 | tail -1)"; done
