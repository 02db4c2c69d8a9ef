#!/bin/bash
# Runs ON THE GPU BOX: config 4 at the round's last build — rocprofv3 kernel trace of its bench command, instruction counters.
export TMPDIR=/tmp
O=gpurun_out/r04n
P=gpurun_out/profiles
mkdir -p $O $P
CMD="python bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 50 --model laikago_soft --envs-per-gpu 8192 --spin-up-steps 0"
timeout 150 rocprofv3 --kernel-trace --stats -d $O/kt -o k -- $CMD > $P/r04n_bench_laikago_soft8192_f64_under_rocprof.json 2> $O/kt.err
DB=$(ls $O/kt/*.db $O/kt/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > $P/r04n_laikago_soft8192_f64_kernel_stats.txt 2>&1
rm -rf $O/kt
head -5 $P/r04n_laikago_soft8192_f64_kernel_stats.txt | cut -c1-170
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/sq_$i -o p -- $CMD --no-events > $O/sq_$i.log 2>&1
done
python tools/pmc_summary.py $O/sq_* > $P/r04n_laikago_soft8192_f64_sq_counters.txt 2>&1
rm -rf $O/sq_*/
cat $P/r04n_laikago_soft8192_f64_sq_counters.txt | cut -c1-150
