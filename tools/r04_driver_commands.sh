#!/bin/bash
# Runs ON THE GPU BOX: the two commands the driver runs at round end, as it runs them (smoke, then the default bench line).
export TMPDIR=/tmp
P=gpurun_out/profiles
mkdir -p $P
T=${TAG:-r04m}
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $P/${T}_smoke.log 2>&1; tail -2 $P/${T}_smoke.log
S=$(date +%s.%N)
timeout 280 python bench.py --gpus 1 --steps 20 --warmup 5 > $P/${T}_bench_driver_command.json 2> $P/${T}_bench_driver_command.err
E=$(date +%s.%N)
echo "bench.py wall: $(python3 -c "print(round($E-$S,1))") s"
python3 - $P/${T}_bench_driver_command.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('%.4g %s  ms_per_step %.5f  roofline %s' % (d['value'], d['unit'], d['ms_per_step'], {k: d['roofline'].get(k) for k in ('achieved','frac','traffic','kernel_ms_avg')}))
print('cpu_baseline', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('kind'), d['cpu_baseline'].get('cores'))
for k in ('substep_fused','one_rank_with_exchange','auto_reset_rate'):
    if k in d and d[k]: print(k, d[k].get('value'))
PY
