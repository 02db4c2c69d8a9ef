#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/c15
mkdir -p $O
timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline"
for c in 1 2 4; do
  TDS_HIP_STEP_MANY_EAGER=1 TDS_HIP_GRAPH_CHAINS=$c $B > $O/bench_ant4096_eager_c$c.json 2>> $O/bench.err
  TDS_HIP_STEP_MANY_EAGER=1 TDS_HIP_GRAPH_CHAINS=$c $B --envs-per-gpu 8192 > $O/bench_ant8192_eager_c$c.json 2>> $O/bench.err
  TDS_HIP_STEP_MANY_EAGER=1 TDS_HIP_GRAPH_CHAINS=$c $B --envs-per-gpu 16384 > $O/bench_ant16384_eager_c$c.json 2>> $O/bench.err
  TDS_HIP_STEP_MANY_EAGER=1 TDS_HIP_GRAPH_CHAINS=$c $B --model pendulum5 --dtype f32 > $O/bench_pendulum5_eager_c$c.json 2>> $O/bench.err
done
$B > $O/bench_ant4096_default.json 2>> $O/bench.err
$B --envs-per-gpu 8192 > $O/bench_ant8192_default.json 2>> $O/bench.err
$B --envs-per-gpu 32768 > $O/bench_ant32768_default.json 2>> $O/bench.err
TDS_HIP_GRAPH_CHAINS=1 $B --envs-per-gpu 32768 > $O/bench_ant32768_c1.json 2>> $O/bench.err
TDS_HIP_GRAPH_CHAINS=3 $B --envs-per-gpu 16384 > $O/bench_ant16384_c3.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.3f us'%(1000*d['ms_per_step']))" 2>&1 | tail -1)"; done
