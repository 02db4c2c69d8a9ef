#!/bin/bash
# Runs ON THE GPU BOX (round 4, first call): (1) the hipStreamWaitValue64 probe behind the new exchange wait,
# (2) the SQ / LDS counters the round-3 review asked for: laikago_soft x 8192 (config 4) and Ant x 8192 (config 5's share)
set -u
OUT=gpurun_out/profiles
W=gpurun_out/prof_r04a
mkdir -p $OUT $W
export TMPDIR=/tmp
timeout 120 tools/ubench/wait_value > $OUT/r04_ubench_wait_value.txt 2>&1
echo "wait_value rc=$?" >> $OUT/r04_ubench_wait_value.txt
cat $OUT/r04_ubench_wait_value.txt
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ3="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for C in "laikago_soft8192|--model laikago_soft --envs-per-gpu 8192|graph" "ant8192|--model ant --envs-per-gpu 8192|loop"; do
  IFS='|' read NAME ARGS FORM <<< "$C"
  i=0
  for CTRS in "$SQ1" "$SQ2" "$SQ3"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $W/sq_${NAME}_$i -o p -- python bench.py $ARGS --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events > $W/sq_${NAME}_$i.log 2>&1
  done
  if [ $FORM = loop ]; then python tools/pmc_loop_summary.py 200 $W/sq_${NAME}_* > $OUT/r04_${NAME}_f64_sq_counters.txt 2>&1
  else python tools/pmc_summary.py $W/sq_${NAME}_* > $OUT/r04_${NAME}_f64_sq_counters.txt 2>&1; fi
  grep -v '^# kernel' $OUT/r04_${NAME}_f64_sq_counters.txt | cut -c1-150
  rm -rf $W/sq_${NAME}_*/
done
