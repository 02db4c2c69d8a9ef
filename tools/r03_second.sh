#!/bin/bash
# Runs ON THE GPU BOX (round 3, second contact): all ring / shard tests, benches of the per-step-record form, kernel
# trace of the one-rank ring exchange.
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
timeout 900 python -m pytest tests/test_rings.py -m gpu -q --timeout 300 -s > $O/pytest_rings.log 2>&1; tail -3 $O/pytest_rings.log
timeout 900 python -m pytest tests/test_shard_two_ranks_one_gpu.py -m gpu -q --timeout 280 > $O/pytest_two.log 2>&1; tail -3 $O/pytest_two.log
timeout 600 python -m pytest tests/test_multi_gpu.py -m gpu -q --timeout 200 > $O/pytest_multi.log 2>&1; tail -2 $O/pytest_multi.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > $O/bench_default20.json 2> $O/bench_default20.err
$B --steps 1000 --warmup 100 > $O/bench_1000.json 2> $O/bench_1000.err
$B --steps 1000 --warmup 100 --records last --no-secondary > $O/bench_1000_last.json 2> $O/bench_1000_last.err
$B --steps 1000 --warmup 100 --force-gather > $O/bench_fg1000.json 2> $O/bench_fg1000.err
$B --steps 20 --warmup 5 --force-gather > $O/bench_fg20.json 2> $O/bench_fg20.err
$B --steps 1000 --warmup 100 --force-gather --shard-graph > $O/bench_fg1000_graph.json 2> $O/bench_fg1000_graph.err
TDS_HIP_RING_NOFENCE=0 $B --steps 1000 --warmup 100 --force-gather > $O/bench_fg1000_fence.json 2> $O/bench_fg1000_fence.err
$B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/bench_8192.json 2> $O/bench_8192.err
$B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 --no-secondary > $O/bench_laikago.json 2> $O/bench_laikago.err
$B --steps 500 --warmup 50 --model pendulum5 --dtype f32 --no-secondary > $O/bench_pendulum5.json 2> $O/bench_pendulum5.err
for V in fg; do
  EX=
  rocprofv3 --kernel-trace --stats -d $O/kt_$V -o k -- python bench.py --no-cpu-baseline --steps 192 --warmup 64 --force-gather --spin-up-steps 0 $EX > $O/kt_$V.log 2>&1
  DB=$(ls $O/kt_$V/*.db $O/kt_$V/*/*.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py "$DB" > $O/kt_${V}_stats.txt 2>&1
  python tools/rocprof_timeline.py "$DB" 120 > $O/kt_${V}_timeline.txt 2>&1
  rm -rf $O/kt_$V
done
for f in $O/bench_*.json; do echo "$f: $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x=['%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step'])]
    for k in ('substep_fused','one_rank_with_exchange','auto_reset_rate'):
        if k in d and d[k]: x.append(k+'='+('%.4g'%d[k]['value'] if 'value' in d[k] else d[k].get('error','?')[:80]))
    x.append(str(d['config'].get('exchange_form')))
    print(' '.join(x))
except Exception as e:
    print('ERR', e)
P
)"; done
head -12 $O/kt_fg_stats.txt | cut -c1-200
