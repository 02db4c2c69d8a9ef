#!/bin/bash
# first GPU call of round 2: full GPU test-suite + the new bench forms
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/c1
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_multi_gpu.py 2>&1 | tail -40 > gpurun_out/c1/pytest_gpu.log
timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q 2>&1 | tail -60 > gpurun_out/c1/pytest_multi.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B > gpurun_out/c1/bench_ant4096_f64_graph.json 2> gpurun_out/c1/bench1.err
$B --no-graph > gpurun_out/c1/bench_ant4096_f64_eager.json 2>> gpurun_out/c1/bench1.err
$B --steps 20 --warmup 5 > gpurun_out/c1/bench_ant4096_f64_graph_20.json 2>> gpurun_out/c1/bench1.err
$B --steps 20 --warmup 5 --no-graph > gpurun_out/c1/bench_ant4096_f64_eager_20.json 2>> gpurun_out/c1/bench1.err
$B --dtype f32 > gpurun_out/c1/bench_ant4096_mixed.json 2>> gpurun_out/c1/bench1.err
$B --dtype f32 --model pendulum5 > gpurun_out/c1/bench_pendulum5_4096_mixed.json 2>> gpurun_out/c1/bench1.err
$B --dtype f32-pure --model pendulum5 > gpurun_out/c1/bench_pendulum5_4096_f32pure.json 2>> gpurun_out/c1/bench1.err
$B --force-gather > gpurun_out/c1/bench_ant4096_shard1_rccl.json 2>> gpurun_out/c1/bench1.err
TDS_BENCH_RCCL_SINGLE=0 $B --force-gather > gpurun_out/c1/bench_ant4096_shard1_copy.json 2>> gpurun_out/c1/bench1.err
$B --envs-per-gpu 8192 --force-gather > gpurun_out/c1/bench_ant8192_shard1_rccl.json 2>> gpurun_out/c1/bench1.err
$B --envs-per-gpu 8192 > gpurun_out/c1/bench_ant8192_f64.json 2>> gpurun_out/c1/bench1.err
timeout 300 python bench.py > gpurun_out/c1/bench_default.json 2>> gpurun_out/c1/bench1.err
tail -5 gpurun_out/c1/pytest_gpu.log; tail -5 gpurun_out/c1/pytest_multi.log
for f in gpurun_out/c1/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print('%.4g'%d['value'], d['ms_per_step'], d.get('pipelined_gather',{}).get('value'))" 2>&1 | tail -1)"; done
tail -3 gpurun_out/c1/bench1.err
