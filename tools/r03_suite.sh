#!/bin/bash
# Runs ON THE GPU BOX (round 3): whole GPU test-suite + the bench lines + phase breakdowns -> gpurun_out/$1/
export TMPDIR=/tmp
O=gpurun_out/${1:-r03x}
mkdir -p $O
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; tail -4 $O/pytest.log
fi
B="timeout 300 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > $O/bench_default20.json 2> $O/bench_default20.err
$B --steps 1000 --warmup 100 > $O/bench_1000.json 2> $O/bench_1000.err
$B --steps 1000 --warmup 100 --records last --no-secondary > $O/bench_1000_last.json 2> $O/bench_1000_last.err
$B --steps 1000 --warmup 100 --no-graph --no-secondary > $O/bench_1000_nograph.json 2> $O/bench_1000_nograph.err
$B --steps 1000 --warmup 100 --force-gather > $O/bench_fg1000.json 2> $O/bench_fg1000.err
$B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/bench_8192.json 2> $O/bench_8192.err
$B --steps 500 --warmup 50 --envs-per-gpu 16384 --no-secondary > $O/bench_16384.json 2> $O/bench_16384.err
$B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 --no-secondary > $O/bench_laikago.json 2> $O/bench_laikago.err
$B --steps 500 --warmup 50 --model pendulum5 --dtype f32 --no-secondary > $O/bench_pendulum5.json 2> $O/bench_pendulum5.err
timeout 200 python tools/profile_phases.py ant 4096 0 100 > $O/phases_ant4096.txt 2>&1
timeout 200 python tools/profile_phases.py laikago_soft 8192 0 100 > $O/phases_laikago_soft8192.txt 2>&1
for f in $O/bench_*.json; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x=['%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step'])]
    for k in ('substep_fused','one_rank_with_exchange','auto_reset_rate'):
        if k in d and d[k]: x.append(k+'='+('%.4g'%d[k]['value'] if 'value' in d[k] else d[k].get('error','?')[:80]))
    x.append(str(d['config'].get('exchange_form'))); x.append('nonfinite=%d'%d['nonfinite_envs'])
    print(' '.join(x))
except Exception as e:
    print('ERR', e)
P
)"; done
head -40 $O/phases_ant4096.txt | tail -32
