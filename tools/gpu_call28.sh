#!/bin/bash
export TMPDIR=/tmp
B="timeout 300 python bench.py --no-cpu-baseline --model pendulum5 --dtype f32"
for c in 1 2; do
  echo "graph chains=$c: $($B --chains $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))")"
  echo "eager chains=$c: $(TDS_HIP_STEP_MANY_EAGER=1 $B --chains $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))")"
  echo "graph chains=$c 8192: $($B --chains $c --envs-per-gpu 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))")"
  echo "graph chains=$c W1: $(TDS_HIP_W2=0 $B --chains $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))")"
done
