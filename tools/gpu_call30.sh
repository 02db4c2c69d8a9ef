#!/bin/bash
export TMPDIR=/tmp
B="timeout 300 python bench.py --no-cpu-baseline"
for a in "--model laikago_soft --envs-per-gpu 8192" "--model laikago_soft --envs-per-gpu 8192 --chains 1" "" "--chains 1" "--envs-per-gpu 8192"; do
  echo "$a: $($B $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.2f us'%(1000*d['ms_per_step']))")"
done
timeout 200 python tools/profile_phases.py laikago_soft 8192 0 100 2>&1 | head -16 | tail -14
