#!/bin/bash
# laikago_soft on the 16-lane kernel by batch size: us per step of the chained graphs (straight-line form) — or, with extra
# bench.py arguments ("--option step_many_loop=1", "--auto-reset", ...), of the form they select.  How the kernel's LDS
# footprint (workgroups per compute unit) shows: 7168 environments = 1792 workgroups = 7 per CU.
#     usage (GPU box): tools/quad_occupancy_sweep.sh [bench.py arguments ...]
for N in ${SIZES:-2048 4096 6144 7168 8192 14336}; do
  timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu $N "$@" > /tmp/q.json 2>/tmp/q.err
  python3 -c "
import json;d=json.load(open('/tmp/q.json'));print($N,'$*','value %.4g us/step %.2f'%(d['value'],1e3*d['ms_per_step']))" || tail -3 /tmp/q.err
done
