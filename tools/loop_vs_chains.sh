#!/bin/bash
# Runs ON THE GPU BOX: tds_hip_step_many as one step-loop launch (TDS_HIP_STEP_MANY_LOOP=1) against the chained graphs
export TMPDIR=/tmp
B="timeout 300 python bench.py --no-cpu-baseline"
for a in "--envs-per-gpu 2048" "" "--envs-per-gpu 8192" "--envs-per-gpu 16384" "--model laikago_soft --envs-per-gpu 8192" "--model laikago_soft --envs-per-gpu 4096" "--dtype f32"; do
  for f in 0 1; do
    echo "[$a] loop=$f: $(TDS_HIP_STEP_MANY_LOOP=$f $B $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']), d['finite'])")"
  done
done
