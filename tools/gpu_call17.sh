#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/c17
mkdir -p $O
timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="timeout 300 python bench.py --no-cpu-baseline"
for c in 1 2 3 4; do
  TDS_HIP_GRAPH_CHAINS=$c $B > $O/bench_ant4096_c$c.json 2>> $O/bench.err
  TDS_HIP_GRAPH_CHAINS=$c $B --envs-per-gpu 8192 > $O/bench_ant8192_c$c.json 2>> $O/bench.err
  TDS_HIP_GRAPH_CHAINS=$c $B --envs-per-gpu 16384 > $O/bench_ant16384_c$c.json 2>> $O/bench.err
  TDS_HIP_GRAPH_CHAINS=$c $B --model pendulum5 --dtype f32 > $O/bench_pendulum5_c$c.json 2>> $O/bench.err
  TDS_HIP_GRAPH_CHAINS=$c $B --model laikago_soft --envs-per-gpu 8192 > $O/bench_laikago_soft8192_c$c.json 2>> $O/bench.err
done
TDS_HIP_W2=0 TDS_HIP_GRAPH_CHAINS=2 $B > $O/bench_ant4096_w1_c2.json 2>> $O/bench.err
TDS_HIP_W2=2 TDS_HIP_GRAPH_CHAINS=2 $B --envs-per-gpu 8192 > $O/bench_ant8192_w2forced_c2.json 2>> $O/bench.err
TDS_HIP_GRAPH_CHAINS=2 $B --steps 20 --warmup 5 > $O/bench_ant4096_c2_20steps.json 2>> $O/bench.err
TDS_HIP_GRAPH_CHAINS=2 $B --envs-per-gpu 2048 > $O/bench_ant2048_c2.json 2>> $O/bench.err
TDS_HIP_GRAPH_CHAINS=1 $B --envs-per-gpu 2048 > $O/bench_ant2048_c1.json 2>> $O/bench.err
for f in $O/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.3f us'%(1000*d['ms_per_step']))" 2>&1 | tail -1)"; done
