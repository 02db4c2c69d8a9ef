#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/c2
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/c2/pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B --force-gather > gpurun_out/c2/bench_ant4096_shard1_rccl_graph.json 2> gpurun_out/c2/bench.err
$B --force-gather --no-graph --pipelined-block 0 > gpurun_out/c2/bench_ant4096_shard1_rccl_eager.json 2>> gpurun_out/c2/bench.err
$B --force-gather --steps 20 --warmup 5 --pipelined-block 0 > gpurun_out/c2/bench_ant4096_shard1_rccl_graph_20.json 2>> gpurun_out/c2/bench.err
$B --force-gather --envs-per-gpu 8192 --pipelined-block 0 > gpurun_out/c2/bench_ant8192_shard1_rccl_graph.json 2>> gpurun_out/c2/bench.err
$B --force-gather --gather-dtype f64 --pipelined-block 0 > gpurun_out/c2/bench_ant4096_shard1_rccl_graph_f64wire.json 2>> gpurun_out/c2/bench.err
python tools/auto_reset_modes.py > gpurun_out/c2/auto_reset_modes_before.txt 2>&1
tail -8 gpurun_out/c2/pytest_gpu.log
for f in gpurun_out/c2/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], d['ms_per_step'], d.get('pipelined_gather',{}).get('value'), d['config']['launch'])" 2>&1 | tail -1)"; done
tail -5 gpurun_out/c2/bench.err; cat gpurun_out/c2/auto_reset_modes_before.txt | tail -9
