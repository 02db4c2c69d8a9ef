#!/bin/bash
# Runs ON THE GPU BOX: the GPU suite + config 4's line at the current build.
export TMPDIR=/tmp
O=gpurun_out/r04s
P=gpurun_out/profiles
mkdir -p $O $P
timeout 500 python -m pytest tests -m gpu -q -n 4 --timeout 120 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-200
cp $O/pytest_gpu.log $P/r04o_pytest_gpu.log
NS="timeout 120 python bench.py --no-cpu-baseline --no-secondary"
$NS --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $P/r04o_bench_laikago_soft8192_f64.json 2> $O/laikago.err
$NS --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $O/laikago2.json 2> $O/laikago2.err
for f in $P/r04o_bench_laikago_soft8192_f64.json $O/laikago2.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('laikago_soft x 8192: %.4g env-steps/s %.2f us/step nonfinite %s'%(d['value'],1000*d['ms_per_step'],d.get('nonfinite_envs')))"; done
