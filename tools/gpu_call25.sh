#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/profiles
mkdir -p $O
timeout 900 python tools/auto_reset_modes.py > $O/r02d_auto_reset_modes.txt 2>&1
timeout 900 python tools/rollout_modes.py > $O/r02d_rollout_modes.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --force-gather > $O/r02d_bench_ant4096_shard1_rccl_graph.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/r02d_bench_ant4096_f64_20steps.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --no-graph > $O/r02d_bench_ant4096_f64_nograph.json 2> /dev/null
tail -12 $O/r02d_auto_reset_modes.txt; tail -14 $O/r02d_rollout_modes.txt
for f in $O/r02d_bench_ant4096_shard1_rccl_graph.json $O/r02d_bench_ant4096_f64_20steps.json $O/r02d_bench_ant4096_f64_nograph.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', '%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))"; done
