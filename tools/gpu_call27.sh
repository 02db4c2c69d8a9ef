#!/bin/bash
export TMPDIR=/tmp
B="timeout 300 python bench.py --no-cpu-baseline"
for q in 4 8; do
for c in 2 3 4; do
  echo "GPU_MAX_HW_QUEUES=$q chains=$c: $(GPU_MAX_HW_QUEUES=$q $B --chains $c 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))")  8192: $(GPU_MAX_HW_QUEUES=$q $B --chains $c --envs-per-gpu 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))")"
done
done
