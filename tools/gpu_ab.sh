#!/bin/bash
# same-box A/B: round-1 final library (ab_r1/, one-wave kernels) vs the current tree, alternating
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for rep in 1 2; do
for n in 4096 8192; do
  (cd ab_r1 && timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu $n 2>/dev/null | tail -1 > ../gpurun_out/ab/r1_ant${n}_$rep.json)
  TDS_HIP_W2=0 timeout 300 python bench.py --no-cpu-baseline --no-graph --envs-per-gpu $n 2>/dev/null | tail -1 > gpurun_out/ab/cur_w1_eager_ant${n}_$rep.json
  TDS_HIP_W2=0 timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu $n 2>/dev/null | tail -1 > gpurun_out/ab/cur_w1_ant${n}_$rep.json
  timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu $n 2>/dev/null | tail -1 > gpurun_out/ab/cur_ant${n}_$rep.json
done
done
for f in gpurun_out/ab/*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.3f us'%(d['roofline']['kernel_ms_avg']*1e3))" 2>&1 | tail -1)"; done
