#!/bin/bash
# Runs ON THE GPU BOX (round 4, seventh call; the sixth was killed by its own limit: the GPU suite hung in
# test_no_step_reads_stale_lds[loop-cartpole] on the no-MachineLICM build).  Measurements first, the hang's diagnosis last,
# everything under short limits.
export TMPDIR=/tmp
O=gpurun_out/r04g
P=gpurun_out/profiles
mkdir -p $O $P
timeout 200 python tools/ab_slots.py --reps 5 > $P/r04_ab_slots_ant4096.txt 2>&1; cat $P/r04_ab_slots_ant4096.txt
B="timeout 120 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > $P/r04b_bench_ant4096_f64_default.json 2> $O/default20.err
$B --steps 1000 --warmup 100 > $P/r04b_bench_ant4096_f64_1000.json 2> $O/b1000.err
NS="$B --no-secondary"
$NS --steps 500 --warmup 50 --envs-per-gpu 8192 > $P/r04b_bench_ant8192_f64.json 2> $O/ant8192.err
$NS --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $P/r04b_bench_laikago_soft8192_f64.json 2> $O/laikago.err
$NS --steps 500 --warmup 50 --model pendulum5 --dtype f32 > $P/r04b_bench_pendulum5_4096_f32rec.json 2> $O/pendulum5.err
TDS_BENCH_TUNE_EXCHANGE=1 $NS --steps 1024 --warmup 128 --force-gather > $P/r04b_bench_ant4096_one_rank_exchange_1024.json 2> $O/fg.err
TDS_BENCH_TUNE_EXCHANGE=1 $NS --steps 20 --warmup 5 --force-gather > $P/r04b_bench_ant4096_one_rank_exchange_20.json 2> $O/fg20.err
for f in $P/r04b_bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x=['%.4g env-steps/s'%d['value'], '%.2f us/step'%(1000*d['ms_per_step'])]
    for k in ('substep_fused','one_rank_with_exchange','auto_reset_rate'):
        if k in d and d[k]: x.append(k+'='+('%.4g'%d[k]['value'] if 'value' in d[k] else d[k].get('error','?')[:80]))
    x.append('form=%s tune=%s'%(d['config'].get('exchange_form'), d['config'].get('exchange_tune')))
    print(' '.join(x))
except Exception as e:
    print('ERR', e)
PY
)"; done | tee $P/r04b_bench_lines.txt
# instruction cache of the headline launch (64 KB per pair of compute units; the two-wavefront loop kernel is 69 KB of code)
i=0
for CTRS in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/ic_$i -o p -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events > $O/ic_$i.log 2>&1
done
python tools/pmc_loop_summary.py 1000 $O/ic_* > $P/r04_ant4096_f64_icache_counters.txt 2>&1
rm -rf $O/ic_*/
grep -v '^# kernel' $P/r04_ant4096_f64_icache_counters.txt | cut -c1-150
timeout 420 python tools/diag_loop_hang.py > $P/r04_diag_loop_hang.txt 2>&1; cat $P/r04_diag_loop_hang.txt | cut -c1-300
