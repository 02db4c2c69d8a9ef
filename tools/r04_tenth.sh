#!/bin/bash
# Runs ON THE GPU BOX (round 4, tenth call): the whole GPU suite on the build with the workgroup constant table (xdist workers,
# per-test limit), then this build's lines of every 1-GPU config and of one rank through the exchange, the rocprofv3 kernel
# trace of the driver's exact command.
export TMPDIR=/tmp
O=gpurun_out/r04l
P=gpurun_out/profiles
mkdir -p $O $P
timeout 420 python -m pytest tests -m gpu -q -n 4 --timeout 90 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log | cut -c1-300
cp $O/pytest_gpu.log $P/r04e_pytest_gpu.log
B="timeout 120 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > $P/r04e_bench_ant4096_f64_default.json 2> $O/default20.err
$B --steps 1000 --warmup 100 > $P/r04e_bench_ant4096_f64_1000.json 2> $O/b1000.err
NS="$B --no-secondary"
$NS --steps 500 --warmup 50 --envs-per-gpu 8192 > $P/r04e_bench_ant8192_f64.json 2> $O/ant8192.err
$NS --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $P/r04e_bench_laikago_soft8192_f64.json 2> $O/laikago.err
$NS --steps 500 --warmup 50 --model pendulum5 --dtype f32 > $P/r04e_bench_pendulum5_4096_f32rec.json 2> $O/pendulum5.err
TDS_BENCH_TUNE_EXCHANGE=1 $NS --steps 1024 --warmup 128 --force-gather > $P/r04e_bench_ant4096_one_rank_exchange_1024.json 2> $O/fg.err
$NS --steps 500 --warmup 50 --model pendulum5 --dtype f64 > $P/r04e_bench_pendulum5_4096_f64.json 2> $O/pendulum5_f64.err
for f in $P/r04e_bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x=['%.4g env-steps/s'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'frac %.4f'%d['roofline']['frac']]
    for k in ('substep_fused','one_rank_with_exchange','auto_reset_rate'):
        if k in d and d[k]: x.append(k+'='+('%.4g'%d[k]['value'] if 'value' in d[k] else d[k].get('error','?')[:80]))
    x.append('form=%s tune=%s'%(d['config'].get('exchange_form'), d['config'].get('exchange_tune')))
    print(' '.join(x))
except Exception as e:
    print('ERR', e)
PY
)"; done | tee $P/r04e_bench_lines.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_def -o k -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_def.log 2>&1
DB=$(ls $O/kt_def/*.db $O/kt_def/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > $P/r04e_ant4096_f64_default_kernel_stats.txt 2>&1
python tools/rocprof_dispatches.py "$DB" > $P/r04e_ant4096_f64_default_dispatches.txt 2>&1
rm -rf $O/kt_def
head -8 $P/r04e_ant4096_f64_default_kernel_stats.txt | cut -c1-170
