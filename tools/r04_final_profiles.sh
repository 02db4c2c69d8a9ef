#!/bin/bash
# Runs ON THE GPU BOX (round 4, final evidence at the round's last build, profiles/r04h_*): the GPU suite; every 1-GPU line;
# rocprofv3 kernel trace of the driver's exact command; HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes) of the driver's
# command and of the 1000-step region; SQ / LDS counters of the headline launch; one rank through the exchange: kernel trace
# and timeline; per-phase cycles of the two-wavefront step; launch-time fit.
export TMPDIR=/tmp
O=gpurun_out/r04r
P=gpurun_out/profiles
mkdir -p $O $P
timeout 500 python -m pytest tests -m gpu -q -n 4 --timeout 120 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-200
cp $O/pytest_gpu.log $P/r04h_pytest_gpu.log
B="timeout 120 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > $P/r04h_bench_ant4096_f64_default.json 2> $O/default20.err
$B --steps 1000 --warmup 100 > $P/r04h_bench_ant4096_f64_1000.json 2> $O/b1000.err
NS="$B --no-secondary"
$NS --steps 500 --warmup 50 --envs-per-gpu 8192 > $P/r04h_bench_ant8192_f64.json 2> $O/ant8192.err
$NS --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $P/r04h_bench_laikago_soft8192_f64.json 2> $O/laikago.err
$NS --steps 500 --warmup 50 --model pendulum5 --dtype f32 > $P/r04h_bench_pendulum5_4096_f32rec.json 2> $O/pendulum5.err
TDS_BENCH_TUNE_EXCHANGE=1 $NS --steps 1024 --warmup 128 --force-gather > $P/r04h_bench_ant4096_one_rank_exchange_1024.json 2> $O/fg.err
for f in $P/r04h_bench_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x=['%.4g env-steps/s'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'kernel_ms_avg %.4f'%d['roofline'].get('kernel_ms_avg',-1), 'frac %.4f'%d['roofline']['frac']]
    for k in ('substep_fused','one_rank_with_exchange','auto_reset_rate'):
        if k in d and d[k]: x.append(k+'='+('%.4g'%d[k]['value'] if 'value' in d[k] else d[k].get('error','?')[:80]))
    x.append('form=%s tune=%s'%(d['config'].get('exchange_form'), d['config'].get('exchange_tune')))
    print(' '.join(x))
except Exception as e:
    print('ERR', e)
PY
)"; done | tee $P/r04h_bench_lines.txt
# kernel trace of the driver's exact command
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_def -o k -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_def.log 2>&1
DB=$(ls $O/kt_def/*.db $O/kt_def/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > $P/r04h_ant4096_f64_default_kernel_stats.txt 2>&1
python tools/rocprof_dispatches.py "$DB" > $P/r04h_ant4096_f64_default_dispatches.txt 2>&1
rm -rf $O/kt_def
head -6 $P/r04h_ant4096_f64_default_kernel_stats.txt | cut -c1-170
# HBM traffic
for C in "20 5" "1000 100"; do
  set -- $C
  i=0
  for CTRS in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/pmc_$1_$i -o p -- python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events > $O/pmc_$1_$i.log 2>&1
  done
  python tools/pmc_loop_summary.py $1 $O/pmc_$1_* > $P/r04h_ant4096_f64_$1_pmc_traffic.txt 2>&1
  rm -rf $O/pmc_$1_*/
done
grep -h -v '^# kernel' $P/r04h_ant4096_f64_*_pmc_traffic.txt | cut -c1-140
# SQ / LDS counters of the headline launch
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ3="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
i=0
for CTRS in "$SQ1" "$SQ2" "$SQ3"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/sq_$i -o p -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events > $O/sq_$i.log 2>&1
done
python tools/pmc_loop_summary.py 1000 $O/sq_* > $P/r04h_ant4096_f64_sq_counters_loop.txt 2>&1
rm -rf $O/sq_*/
grep -v '^# kernel' $P/r04h_ant4096_f64_sq_counters_loop.txt | cut -c1-140
# one rank through the exchange: kernel trace + timeline
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_fg -o k -- python bench.py --no-cpu-baseline --no-secondary --steps 512 --warmup 256 --force-gather --spin-up-steps 0 > $O/kt_fg.log 2>&1
DB=$(ls $O/kt_fg/*.db $O/kt_fg/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > $P/r04h_one_rank_exchange_kernel_stats.txt 2>&1
python tools/rocprof_timeline.py "$DB" 60 > $P/r04h_one_rank_exchange_timeline.txt 2>&1
rm -rf $O/kt_fg
head -10 $P/r04h_one_rank_exchange_kernel_stats.txt | cut -c1-170
# phases of the two-wavefront step; launch-time fit
timeout 120 python tools/profile_phases.py ant 4096 > $P/r04h_ant4096_f64_phases.txt 2>&1; tail -32 $P/r04h_ant4096_f64_phases.txt | cut -c1-150
timeout 120 python tools/launch_fit.py > $P/r04h_launch_fit.txt 2>&1; tail -6 $P/r04h_launch_fit.txt | cut -c1-200
