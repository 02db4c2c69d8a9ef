#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/c18
mkdir -p $O
timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B > $O/bench_ant4096.json 2>> $O/bench.err
$B --steps 20 --warmup 5 > $O/bench_ant4096_20steps.json 2>> $O/bench.err
$B --envs-per-gpu 8192 > $O/bench_ant8192.json 2>> $O/bench.err
$B --envs-per-gpu 16384 > $O/bench_ant16384.json 2>> $O/bench.err
$B --envs-per-gpu 32768 > $O/bench_ant32768.json 2>> $O/bench.err
$B --model pendulum5 --dtype f32 > $O/bench_pendulum5.json 2>> $O/bench.err
$B --model laikago_soft --envs-per-gpu 8192 > $O/bench_laikago_soft8192.json 2>> $O/bench.err
$B --model cartpole --envs-per-gpu 64 > $O/bench_cartpole64.json 2>> $O/bench.err
$B --dtype f32 > $O/bench_ant4096_mixed.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.3f us'%(1000*d['ms_per_step']), d['config']['launch'])" 2>&1 | tail -1)"; done
