#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench line, rocprofv3 kernel-trace summary, PMC passes and the
# phase breakdown of the current build.  usage: tools/collect_profiles.sh <tag>   -> gpurun_out/profiles/<tag>_*
set -u
TAG=${1:-rXX}
OUT=gpurun_out/profiles
mkdir -p $OUT gpurun_out/prof_$TAG
export TMPDIR=/tmp
BENCH="python bench.py --steps 500 --warmup 50"
$BENCH > $OUT/${TAG}_bench_ant4096_f64.json 2> gpurun_out/prof_$TAG/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG/kt -o ant -- $BENCH --no-cpu-baseline > gpurun_out/prof_$TAG/kt.log 2>&1
DB=$(ls gpurun_out/prof_$TAG/kt/*.db gpurun_out/prof_$TAG/kt/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > $OUT/${TAG}_ant4096_f64_kernel_stats.txt 2>&1
i=0
for CTRS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d gpurun_out/prof_$TAG/pmc$i -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$TAG/pmc$i.log 2>&1
done
python tools/pmc_summary.py gpurun_out/prof_$TAG/pmc* > $OUT/${TAG}_ant4096_f64_pmc_counters.txt 2>&1
python tools/profile_phases.py ant 4096 > $OUT/${TAG}_ant4096_f64_phases.txt 2>/dev/null
python tools/profile_phases.py laikago 8192 > $OUT/${TAG}_laikago8192_f64_phases.txt 2>/dev/null
python bench.py --steps 300 --warmup 30 --model laikago --envs-per-gpu 8192 --no-cpu-baseline > $OUT/${TAG}_bench_laikago8192_f64.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --envs-per-gpu 16384 --no-cpu-baseline > $OUT/${TAG}_bench_ant16384_f64.json 2>/dev/null
python bench.py --steps 300 --warmup 30 --dtype f32 --no-cpu-baseline > $OUT/${TAG}_bench_ant4096_f32.json 2>/dev/null
tail -n 3 $OUT/${TAG}_ant4096_f64_kernel_stats.txt; cat $OUT/${TAG}_ant4096_f64_pmc_counters.txt | head -30
