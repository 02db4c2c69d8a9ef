#!/bin/bash
# Runs ON THE GPU BOX: the other launch forms of the final build (auto-reset, rollouts, one rank through the shard layer,
# the driver-shaped 20-step run, one launch per step) -> gpurun_out/profiles/<tag>_*.   usage: tools/collect_other_forms.sh [tag]
TAG=${1:-r02d}
export TMPDIR=/tmp
O=gpurun_out/profiles
mkdir -p $O
timeout 900 python tools/auto_reset_modes.py > $O/${TAG}_auto_reset_modes.txt 2>&1
timeout 900 python tools/rollout_modes.py > $O/${TAG}_rollout_modes.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --force-gather > $O/${TAG}_bench_ant4096_shard1_rccl_graph.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/${TAG}_bench_ant4096_f64_20steps.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --no-graph > $O/${TAG}_bench_ant4096_f64_nograph.json 2> /dev/null
timeout 300 python bench.py --no-cpu-baseline --auto-reset > $O/${TAG}_bench_ant4096_f64_auto_reset.json 2> /dev/null
tail -20 $O/${TAG}_auto_reset_modes.txt; tail -14 $O/${TAG}_rollout_modes.txt
for f in $O/${TAG}_bench_ant4096_shard1_rccl_graph.json $O/${TAG}_bench_ant4096_f64_20steps.json $O/${TAG}_bench_ant4096_f64_nograph.json $O/${TAG}_bench_ant4096_f64_auto_reset.json; do python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', '%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))"; done
