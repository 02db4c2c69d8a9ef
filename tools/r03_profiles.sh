#!/bin/bash
# Runs ON THE GPU BOX (round 3): the evidence behind the bench lines -> gpurun_out/profiles/<tag>_*
#   * the driver's exact command (python bench.py --steps 20 --warmup 5): bench line, rocprofv3 kernel trace summary +
#     per-dispatch listing, HBM traffic of the timed launch from separate FETCH_SIZE / WRITE_SIZE passes
#   * the same for a 1000-step region, for Ant x 8192 (config 5's per-GPU share), pendulum5 (config 2), laikago_soft (config 4)
#   * SQ / instruction-cache counters of the headline launch forms
set -u
TAG=${1:-r03}
OUT=gpurun_out/profiles
W=gpurun_out/prof_$TAG
mkdir -p $OUT $W
export TMPDIR=/tmp
rocprofv3 --list-avail > $W/avail.txt 2>&1
grep -i -E "icache|ICACHE|SQ_INSTS_|SQ_WAVE|SQ_BUSY|SQ_WAIT|FETCH_SIZE|WRITE_SIZE|SQ_IFETCH|SQC_" $W/avail.txt | cut -c1-120 | sort -u | head -80 > $OUT/${TAG}_available_counters.txt
# name | bench arguments | steps | warmup
CONFIGS=(
  "ant4096_f64_default|--model ant --envs-per-gpu 4096|20|5"
  "ant4096_f64_1000|--model ant --envs-per-gpu 4096|1000|100"
  "ant8192_f64|--model ant --envs-per-gpu 8192|500|50"
  "pendulum5_4096_f32rec|--model pendulum5 --envs-per-gpu 4096 --dtype f32|500|50"
  "laikago_soft8192_f64|--model laikago_soft --envs-per-gpu 8192|500|50"
)
for C in "${CONFIGS[@]}"; do
  IFS='|' read NAME ARGS K WU <<< "$C"
  python bench.py $ARGS --steps $K --warmup $WU $( [ "$NAME" = ant4096_f64_default ] || echo --no-cpu-baseline ) > $OUT/${TAG}_bench_$NAME.json 2> $W/bench_$NAME.err
  rocprofv3 --kernel-trace --stats -d $W/kt_$NAME -o k -- python bench.py $ARGS --steps $K --warmup $WU --no-cpu-baseline > $W/kt_$NAME.log 2>&1
  DB=$(ls $W/kt_$NAME/*.db $W/kt_$NAME/*/*.db 2>/dev/null | head -1)
  python tools/rocprof_summary.py "$DB" > $OUT/${TAG}_${NAME}_kernel_stats.txt 2>&1
  python tools/rocprof_dispatches.py "$DB" 60 >> $OUT/${TAG}_${NAME}_kernel_stats.txt 2>&1
  rm -rf $W/kt_$NAME
  # HBM traffic of the timed launch: the same launch, nothing else of its build in the process (no spin-up, no
  # secondary measurements), one counter per pass
  i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $W/pmc_${NAME}_$i -o p -- python bench.py $ARGS --steps $K --warmup $WU --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events > $W/pmc_${NAME}_$i.log 2>&1
  done
  python tools/pmc_loop_summary.py $K $W/pmc_${NAME}_* > $OUT/${TAG}_${NAME}_pmc_traffic.txt 2>&1
done
# SQ + instruction cache counters: the headline step-loop launch (rings) and the two-wavefront single-step kernel
for FORM in loop w2; do
  EX=$( [ $FORM = w2 ] && echo --no-graph )
  i=0
  for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
              "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
              "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_IFETCH"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $W/sq_${FORM}_$i -o p -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events $EX > $W/sq_${FORM}_$i.log 2>&1
  done
  if [ $FORM = loop ]; then python tools/pmc_loop_summary.py 200 $W/sq_${FORM}_* > $OUT/${TAG}_ant4096_f64_sq_counters_$FORM.txt 2>&1
  else python tools/pmc_summary.py $W/sq_${FORM}_* > $OUT/${TAG}_ant4096_f64_sq_counters_$FORM.txt 2>&1; fi
done
python tools/profile_phases.py ant 4096 > $OUT/${TAG}_ant4096_f64_phases.txt 2>/dev/null
python tools/profile_phases.py laikago_soft 8192 > $OUT/${TAG}_laikago_soft8192_f64_phases.txt 2>/dev/null
for C in "${CONFIGS[@]}"; do IFS='|' read NAME ARGS K WU <<< "$C"; echo "== $NAME"; head -5 $OUT/${TAG}_${NAME}_kernel_stats.txt | cut -c1-170; cat $OUT/${TAG}_${NAME}_pmc_traffic.txt | cut -c1-170; done
cat $OUT/${TAG}_ant4096_f64_sq_counters_loop.txt | cut -c1-170
tail -30 $OUT/${TAG}_ant4096_f64_sq_counters_w2.txt | cut -c1-170
