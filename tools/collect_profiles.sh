#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): for every 1-GPU BASELINE config — bench line, rocprofv3 kernel-trace summary of
# the SAME command, HBM traffic from separate FETCH_SIZE / WRITE_SIZE PMC passes — plus SQ counters and the phase
# breakdown for the headline config.   usage: tools/collect_profiles.sh <tag>   -> gpurun_out/profiles/<tag>_*
set -u
TAG=${1:-rXX}
OUT=gpurun_out/profiles
W=gpurun_out/prof_$TAG
mkdir -p $OUT $W
export TMPDIR=/tmp
# name | bench arguments
CONFIGS=(
  "ant4096_f64|--model ant --envs-per-gpu 4096"
  "ant8192_f64|--model ant --envs-per-gpu 8192"
  "pendulum5_4096_f32rec|--model pendulum5 --envs-per-gpu 4096 --dtype f32"
  "laikago_soft8192_f64|--model laikago_soft --envs-per-gpu 8192"
)
for C in "${CONFIGS[@]}"; do
  NAME=${C%%|*}; ARGS=${C##*|}
  python bench.py $ARGS --steps 500 --warmup 50 $( [ "$NAME" = ant4096_f64 ] || echo --no-cpu-baseline ) > $OUT/${TAG}_bench_$NAME.json 2> $W/bench_$NAME.err
  rocprofv3 --kernel-trace --stats -d $W/kt_$NAME -o k -- python bench.py $ARGS --steps 500 --warmup 50 --no-cpu-baseline > $W/kt_$NAME.log 2>&1
  DB=$(ls $W/kt_$NAME/*.db $W/kt_$NAME/*/*.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py "$DB" > $OUT/${TAG}_${NAME}_kernel_stats.txt 2>&1
  i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $W/pmc_${NAME}_$i -o p -- python bench.py $ARGS --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $W/pmc_${NAME}_$i.log 2>&1
  done
  python tools/pmc_summary.py $W/pmc_${NAME}_* > $OUT/${TAG}_${NAME}_pmc_counters.txt 2>&1
  # configs whose K steps run as ONE step-loop launch: HBM traffic of that launch (same two counters, same command as the bench line)
  if grep -q "step-loop kernel" $OUT/${TAG}_bench_$NAME.json; then
    i=0
    for CTRS in "FETCH_SIZE" "WRITE_SIZE"; do
      i=$((i+1))
      rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $W/pmcloop_${NAME}_$i -o p -- python bench.py $ARGS --steps 500 --warmup 50 --no-cpu-baseline > $W/pmcloop_${NAME}_$i.log 2>&1
    done
    python tools/pmc_loop_summary.py 500 $W/pmcloop_${NAME}_* > $OUT/${TAG}_${NAME}_pmc_counters_loop.txt 2>&1
  fi
done
# (QUICK=1: bench lines, kernel-trace summaries, HBM traffic and phase breakdowns only)
# SQ counters of the headline kernel (both workgroup forms)
for FORM in $( [ "${QUICK:-0}" = 1 ] || echo w2 w1 ); do
  i=0
  for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
              "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
              "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    TDS_HIP_W2=$( [ $FORM = w2 ] && echo 1 || echo 0 ) rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $W/sq_${FORM}_$i -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph > $W/sq_${FORM}_$i.log 2>&1
  done
  python tools/pmc_summary.py $W/sq_${FORM}_* > $OUT/${TAG}_ant4096_f64_sq_counters_$FORM.txt 2>&1
done
python tools/profile_phases.py ant 4096 > $OUT/${TAG}_ant4096_f64_phases.txt 2>/dev/null
python tools/profile_phases.py laikago_soft 8192 > $OUT/${TAG}_laikago_soft8192_f64_phases.txt 2>/dev/null
python tools/w2_tail.py > $OUT/${TAG}_ant4096_f64_workgroup_times.txt 2>/dev/null
# launch forms of tds_hip_step_many: chained graphs by chain count, and the one-launch step-loop form
[ "${QUICK:-0}" = 1 ] || bash tools/step_many_forms.sh > $OUT/${TAG}_graph_chains.txt 2>/dev/null
# micro-benchmarks (tools/ubench, built in-tree)
for U in $( [ "${QUICK:-0}" = 1 ] || echo launch_clock kernel_boundary two_streams mfma_f64_4x4x4 ); do
  [ -x tools/ubench/$U ] && timeout 120 tools/ubench/$U > $OUT/${TAG}_ubench_$U.txt 2>&1
done
for C in "${CONFIGS[@]}"; do NAME=${C%%|*}; echo "== $NAME"; tail -n 4 $OUT/${TAG}_${NAME}_kernel_stats.txt; cat $OUT/${TAG}_${NAME}_pmc_counters.txt; done
[ "${QUICK:-0}" = 1 ] || head -30 $OUT/${TAG}_ant4096_f64_sq_counters_w2.txt
