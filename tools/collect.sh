#!/bin/bash
# The ONE evidence collector (runs ON THE GPU BOX through gpurun; replaces the per-round one-off batches):
#     gpurun -- 'bash tools/collect.sh <tag> <section> [<section> ...]'      ->  gpurun_out/profiles/<tag>_*
# sections:
#   suite      the GPU test suite                                   <tag>_pytest_gpu.log
#   lines      every 1-GPU bench line (driver's command, 1000 steps, configs 5 / 2 / 4)   <tag>_bench_*.json, <tag>_bench_lines.txt
#   trace      rocprofv3 --kernel-trace --stats of the driver's exact command             <tag>_ant4096_f64_default_{kernel_stats,dispatches}.txt
#   traffic    HBM bytes: separate FETCH_SIZE / WRITE_SIZE passes, 20- and 1000-step      <tag>_ant4096_f64_{20,1000}_pmc_traffic.txt
#   sq         SQ / LDS counters of the headline launch (three passes)                    <tag>_ant4096_f64_sq_counters_loop.txt
#   others     the same two for config 5's share (Ant x 8192), config 4 (laikago_soft x 8192, +-0.4 actions with auto-reset: its
#              default line; and the 16-lane kernel's step-loop form at 4096) and config 2 (pendulum5, float records):
#              traffic + SQ counters of THEIR kernels (tds_oct_kernel, tds_quad_kernel, tds_chain_kernel)
#   config2    traffic + SQ counters + kernel trace of config 2 (pendulum5 x 4096, float records: tds_chain_kernel); config 1's line
#   exchange   one rank through tds_hip_shard_step_many (0 and 7 loopback peers): lines, kernel trace, timeline
#   tworank    TWO processes on the one GPU through the peer-store exchange, rocprofv3 kernel trace of each + timeline
#   phases     per-phase cycles of the two-wavefront step (both wavefronts), Ant and Laikago
#   resources  (no GPU needed) compiler-reported registers / scratch / SGPR spills of the f64 plain kernels
# Copy what is to be judged from gpurun_out/profiles/ into profiles/ (tracked).
set -u
export TMPDIR=/tmp
TAG=${1:?tag}; shift
O=gpurun_out/work_$TAG
P=gpurun_out/profiles
mkdir -p $O $P
B="timeout 300 python bench.py --no-cpu-baseline"
NS="$B --no-secondary"
db_of() { ls $1/*.db $1/*/*.db 2>/dev/null | head -1; }
line_of() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x=['%.4g env-steps/s'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'kernel_ms_avg %.4f'%d['roofline'].get('kernel_ms_avg',-1), 'frac %.4f'%d['roofline']['frac']]
    for k in ('steady_state_1000','substep_fused','one_rank_with_exchange','one_rank_with_exchange_7_loopback_peers','auto_reset_rate','no_auto_reset_rate'):
        if k in d and d[k]: x.append(k+'='+('%.4g'%d[k]['value'] if 'value' in d[k] else d[k].get('error','?')[:80]))
    x.append('form=%s'%(d['config'].get('exchange_form')))
    print(' '.join(x))
except Exception as e:
    print('ERR', e)
PY
}
for S in "$@"; do case $S in
suite)
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $P/${TAG}_pytest_gpu.log 2>&1
  tail -3 $P/${TAG}_pytest_gpu.log | cut -c1-200 ;;
lines)
  timeout 600 python bench.py --steps 20 --warmup 5 > $P/${TAG}_bench_ant4096_f64_default.json 2> $O/default20.err
  $B --steps 1000 --warmup 100 > $P/${TAG}_bench_ant4096_f64_1000.json 2> $O/b1000.err
  $NS --steps 500 --warmup 50 --envs-per-gpu 8192 > $P/${TAG}_bench_ant8192_f64.json 2> $O/ant8192.err
  $B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $P/${TAG}_bench_laikago_soft8192_f64.json 2> $O/laikago.err
  $B --steps 1000 --warmup 100 --model laikago_soft --envs-per-gpu 4096 > $P/${TAG}_bench_laikago_soft4096_f64.json 2> $O/laikago4096.err
  $NS --steps 500 --warmup 50 --model pendulum5 --dtype f32 > $P/${TAG}_bench_pendulum5_4096_f32rec.json 2> $O/pendulum5.err
  for f in $P/${TAG}_bench_*.json; do echo "$(basename $f .json): $(line_of $f)"; done | tee $P/${TAG}_bench_lines.txt ;;
trace)
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_def -o k -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_def.log 2>&1
  DB=$(db_of $O/kt_def)
  python tools/rocprof_summary.py "$DB" > $P/${TAG}_ant4096_f64_default_kernel_stats.txt 2>&1
  python tools/rocprof_dispatches.py "$DB" > $P/${TAG}_ant4096_f64_default_dispatches.txt 2>&1
  rm -rf $O/kt_def
  head -6 $P/${TAG}_ant4096_f64_default_kernel_stats.txt | cut -c1-170 ;;
traffic)
  for C in "20 5" "1000 100"; do
    set -- $C
    i=0
    for CTRS in FETCH_SIZE WRITE_SIZE; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/pmc_$1_$i -o p -- python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events > $O/pmc_$1_$i.log 2>&1
    done
    python tools/pmc_loop_summary.py $1 $O/pmc_$1_* > $P/${TAG}_ant4096_f64_$1_pmc_traffic.txt 2>&1
    rm -rf $O/pmc_$1_*/
  done
  grep -h -v '^# kernel' $P/${TAG}_ant4096_f64_*_pmc_traffic.txt | cut -c1-140 ;;
sq)
  SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
  SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  SQ3="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
  i=0
  for CTRS in "$SQ1" "$SQ2" "$SQ3"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/sq_$i -o p -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events > $O/sq_$i.log 2>&1
  done
  python tools/pmc_loop_summary.py 1000 $O/sq_* > $P/${TAG}_ant4096_f64_sq_counters_loop.txt 2>&1
  rm -rf $O/sq_*/
  grep -v '^# kernel' $P/${TAG}_ant4096_f64_sq_counters_loop.txt | cut -c1-140 ;;
others)
  SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
  SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  SQ3="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
  # name | steps of the launch the counters are read from (0: single-step launches, mean) | bench.py arguments
  #   laikago_soft8192_loop: config 4 in the 16-lane kernel's step-loop form, wide workgroups (its default since r06e), one 500-step
  #                          launch (no reset, +-0.1 rad)
  #   laikago_soft8192_single: the single-step launches 8192 ran until r06d (option quad_wide = 0; +-0.4 rad, auto-reset)
  #   laikago_soft4096_loop: the step-loop form in one-wavefront workgroups, one 500-step launch (no reset, +-0.1 rad)
  while read -r NAME K ARGS; do
    i=0
    for CTRS in FETCH_SIZE WRITE_SIZE "$SQ1" "$SQ2" "$SQ3"; do
      i=$((i+1))
      timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/o_${NAME}_$i -o p -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events $ARGS > $O/o_${NAME}_$i.log 2>&1
    done
    python tools/pmc_loop_summary.py $K $O/o_${NAME}_1 $O/o_${NAME}_2 > $P/${TAG}_${NAME}_pmc_traffic.txt 2>&1
    python tools/pmc_loop_summary.py $K $O/o_${NAME}_3 $O/o_${NAME}_4 $O/o_${NAME}_5 > $P/${TAG}_${NAME}_sq_counters.txt 2>&1
    rm -rf $O/o_${NAME}_*/
    echo "$NAME:"; grep -h -v '^# kernel' $P/${TAG}_${NAME}_pmc_traffic.txt | cut -c1-140
  done <<'CFGS'
ant8192_f64 500 --envs-per-gpu 8192
laikago_soft8192_f64_loop 500 --model laikago_soft --envs-per-gpu 8192 --no-auto-reset
laikago_soft8192_f64_single 0 --model laikago_soft --envs-per-gpu 8192 --option quad_wide=0
laikago_soft4096_f64_loop 500 --model laikago_soft --envs-per-gpu 4096 --no-auto-reset
pendulum5_4096_f32rec 500 --model pendulum5 --dtype f32
CFGS
  ;;
config2)
  SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
  SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
  SQ3="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
  NAME=pendulum5_4096_f32rec
  i=0
  for CTRS in FETCH_SIZE WRITE_SIZE "$SQ1" "$SQ2" "$SQ3"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/o_${NAME}_$i -o p -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events --model pendulum5 --dtype f32 > $O/o_${NAME}_$i.log 2>&1
  done
  python tools/pmc_loop_summary.py 500 $O/o_${NAME}_1 $O/o_${NAME}_2 > $P/${TAG}_${NAME}_pmc_traffic.txt 2>&1
  python tools/pmc_loop_summary.py 500 $O/o_${NAME}_3 $O/o_${NAME}_4 $O/o_${NAME}_5 > $P/${TAG}_${NAME}_sq_counters.txt 2>&1
  rm -rf $O/o_${NAME}_*/
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_p5 -o k -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-secondary --model pendulum5 --dtype f32 > $O/kt_p5.log 2>&1
  python tools/rocprof_summary.py "$(db_of $O/kt_p5)" > $P/${TAG}_${NAME}_kernel_stats.txt 2>&1
  rm -rf $O/kt_p5
  $B --steps 500 --warmup 50 --model pendulum5 --dtype f32 > $P/${TAG}_bench_pendulum5_4096_f32rec.json 2> $O/pendulum5.err
  $B --steps 500 --warmup 50 --model cartpole --envs-per-gpu 64 > $P/${TAG}_bench_cartpole64_f64.json 2> $O/cartpole.err
  grep -h -v '^# kernel' $P/${TAG}_${NAME}_pmc_traffic.txt | cut -c1-140; head -4 $P/${TAG}_${NAME}_kernel_stats.txt | cut -c1-160
  echo "cartpole x 64: $(line_of $P/${TAG}_bench_cartpole64_f64.json)" ;;
exchange)
  for LB in 0 7; do
    $NS --steps 1024 --warmup 256 --force-gather --option shard_peer_loopback=$LB > $P/${TAG}_bench_ant4096_one_rank_exchange_${LB}peers_1024.json 2> $O/fg$LB.err
    echo "one rank, $LB loopback peers, 1024 steps: $(line_of $P/${TAG}_bench_ant4096_one_rank_exchange_${LB}peers_1024.json)"
  done | tee $P/${TAG}_one_rank_exchange_lines.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_fg -o k -- python bench.py --no-cpu-baseline --no-secondary --steps 512 --warmup 256 --force-gather --spin-up-steps 0 --option shard_peer_loopback=7 > $O/kt_fg.log 2>&1
  DB=$(db_of $O/kt_fg)
  python tools/rocprof_summary.py "$DB" > $P/${TAG}_one_rank_exchange_kernel_stats.txt 2>&1
  python tools/rocprof_timeline.py "$DB" 60 > $P/${TAG}_one_rank_exchange_timeline.txt 2>&1
  rm -rf $O/kt_fg ;;
tworank)
  bash tools/two_rank_peer_trace.sh $TAG $TAG ;;
phases)
  python tools/profile_phases.py ant 4096 > $P/${TAG}_ant4096_f64_phases.txt 2>/dev/null
  python tools/profile_phases.py laikago_soft 8192 > $P/${TAG}_laikago_soft8192_f64_phases.txt 2>/dev/null
  tail -30 $P/${TAG}_ant4096_f64_phases.txt | cut -c1-160 ;;
resources)
  bash tools/kernel_resources.sh f64 0 > $P/${TAG}_kernel_resources_f64_k0.txt 2>&1
  head -40 $P/${TAG}_kernel_resources_f64_k0.txt | cut -c1-140 ;;
*) echo "unknown section $S" ;;
esac; done
