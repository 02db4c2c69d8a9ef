#!/bin/bash
# Runs ON THE GPU BOX: first contact of the per-step record rings + ring exchange with hardware (round 3).
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout 900 python -m pytest tests/test_rings.py -m gpu -q -x --timeout 300 -s > $O/pytest_rings.log 2>&1; tail -5 $O/pytest_rings.log
timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q --timeout 200 > $O/pytest_multi.log 2>&1; tail -5 $O/pytest_multi.log
timeout 900 python -m pytest tests/test_shard_two_ranks_one_gpu.py -m gpu -q --timeout 280 > $O/pytest_two.log 2>&1; tail -5 $O/pytest_two.log
B="timeout 300 python bench.py"
$B --steps 20 --warmup 5 > $O/bench_default20.json 2> $O/bench_default20.err
$B --steps 1000 --warmup 100 --no-cpu-baseline > $O/bench_1000.json 2> $O/bench_1000.err
$B --steps 1000 --warmup 100 --no-cpu-baseline --records last --no-secondary > $O/bench_1000_last.json 2> $O/bench_1000_last.err
$B --steps 20 --warmup 5 --no-cpu-baseline --force-gather > $O/bench_force_gather20.json 2> $O/bench_force_gather20.err
$B --steps 1000 --warmup 100 --no-cpu-baseline --force-gather > $O/bench_force_gather1000.json 2> $O/bench_force_gather1000.err
$B --steps 1000 --warmup 100 --no-cpu-baseline --force-gather --shard-eager > $O/bench_force_gather1000_eager.json 2> $O/bench_force_gather1000_eager.err
$B --steps 500 --warmup 50 --no-cpu-baseline --envs-per-gpu 8192 > $O/bench_8192.json 2> $O/bench_8192.err
$B --steps 500 --warmup 50 --no-cpu-baseline --model laikago_soft --envs-per-gpu 8192 --no-secondary > $O/bench_laikago.json 2> $O/bench_laikago.err
$B --steps 500 --warmup 50 --no-cpu-baseline --model pendulum5 --dtype f32 --no-secondary > $O/bench_pendulum5.json 2> $O/bench_pendulum5.err
for f in $O/bench_*.json; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x=['%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step'])]
    for k in ('substep_fused','one_rank_with_exchange','auto_reset_rate'):
        if k in d and d[k]: x.append(k+'='+('%.4g'%d[k]['value'] if 'value' in d[k] else d[k].get('error','?')[:80]))
    x.append(str(d['config'].get('exchange_form')))
    x.append('nonfinite=%d'%d['nonfinite_envs'])
    print(' '.join(x))
except Exception as e:
    print('ERR', e)
P
)"; done
tail -3 $O/*.err | cut -c1-300
