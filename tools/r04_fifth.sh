#!/bin/bash
# Runs ON THE GPU BOX (round 4, fifth call — the outputs of calls one to four were lost with their container before they
# were committed): the GPU suite, the driver's line, the steady-state lines of configs 3 / 4 / 5-share, the one-rank
# exchange in its forms, the scratch-free (no MachineLICM) step-loop build beside the ordinary one on the same box, HBM
# traffic of the driver's command, SQ / LDS counters of config 4 and of Ant x 8192, kernel traces, the C++ class's rates.
export TMPDIR=/tmp
O=gpurun_out/r04e
P=gpurun_out/profiles
mkdir -p $O $P
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
cp $O/pytest_gpu.log $P/r04_pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline"
V=$PWD/tiny-differentiable-simulator_amd/libtds_hip_xnolicm.so
$B --steps 20 --warmup 5 > $P/r04_bench_ant4096_f64_default.json 2> $O/default20.err
$B --steps 1000 --warmup 100 > $P/r04_bench_ant4096_f64_1000.json 2> $O/b1000.err
NS="$B --no-secondary"
for rep in 1 2; do
  for L in base nolicm; do
    [ $L = base ] && unset TDS_HIP_LIB || export TDS_HIP_LIB=$V
    $NS --steps 1000 --warmup 100 > $O/ab_${L}_ant4096_1000_$rep.json 2> $O/ab_${L}_ant4096_1000_$rep.err
    $NS --steps 20 --warmup 5 > $O/ab_${L}_ant4096_20_$rep.json 2> $O/ab_${L}_ant4096_20_$rep.err
    [ $rep = 1 ] && $NS --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/ab_${L}_ant8192_$rep.json 2> $O/ab_${L}_ant8192_$rep.err
    [ $rep = 1 ] && $NS --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $O/ab_${L}_laikago_$rep.json 2> $O/ab_${L}_laikago_$rep.err
  done
done
unset TDS_HIP_LIB
FG="$NS --steps 1024 --warmup 128 --force-gather"
$FG > $O/fg_rccl_default.json 2> $O/fg_rccl_default.err
$FG --option exchange_w2=1 --option shard_chunk=256 > $O/fg_rccl_w2_c256.json 2> $O/fg_rccl_w2_c256.err
$FG --option shard_wait=0 > $O/fg_rccl_waitkernel.json 2> $O/fg_rccl_waitkernel.err
TDS_BENCH_RCCL_SINGLE=0 $FG > $O/fg_nocomm_default.json 2> $O/fg_nocomm_default.err
TDS_BENCH_RCCL_SINGLE=0 $FG --option exchange_w2=1 --option shard_chunk=256 > $O/fg_nocomm_w2_c256.json 2> $O/fg_nocomm_w2_c256.err
TDS_HIP_LIB=$V $FG > $O/fg_rccl_default_nolicm.json 2> $O/fg_rccl_default_nolicm.err
TDS_HIP_LIB=$V $FG --option exchange_w2=1 --option shard_chunk=256 > $O/fg_rccl_w2_c256_nolicm.json 2> $O/fg_rccl_w2_c256_nolicm.err
$FG --envs-per-gpu 8192 --steps 512 > $O/fg_rccl_8192.json 2> $O/fg_rccl_8192.err
{
for f in $O/ab_*.json $O/fg_*.json; do echo "$(basename $f .json): $(python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%.4g env-steps/s'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'kernel_ms_avg=%.4f'%d['roofline'].get('kernel_ms_avg',-1), 'form=%s'%d['config'].get('exchange_form'), 'nonfinite=%s'%d.get('nonfinite_envs'))
except Exception as e:
    print('ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
PY
)"; done
} > $P/r04_same_box_ab_and_exchange_forms.txt
cat $P/r04_same_box_ab_and_exchange_forms.txt
# kernel trace of the driver's exact command
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_def -o k -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_def.log 2>&1
DB=$(ls $O/kt_def/*.db $O/kt_def/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > $P/r04_ant4096_f64_default_kernel_stats.txt 2>&1
rm -rf $O/kt_def
head -8 $P/r04_ant4096_f64_default_kernel_stats.txt | cut -c1-160
# HBM traffic of the driver's command and the 1000-step region (line-padded y records = the default; packed beside it at 20)
for C in "20 5 line" "1000 100 line"; do
  set -- $C
  i=0
  for CTRS in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/pmc_$3_$1_$i -o p -- python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events --y-stride $3 > $O/pmc_$3_$1_$i.log 2>&1
  done
  python tools/pmc_loop_summary.py $1 $O/pmc_$3_$1_* > $P/r04_ant4096_f64_$1_ystride_$3_pmc_traffic.txt 2>&1
  rm -rf $O/pmc_$3_$1_*/
done
grep -h -v '^# kernel' $P/r04_ant4096_f64_*_pmc_traffic.txt | cut -c1-140
# SQ / LDS counters: config 4 (straight-line launches from graphs) and Ant x 8192 (one-wave step-loop launch)
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ3="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for C in "laikago_soft8192|--model laikago_soft --envs-per-gpu 8192|graph" "ant8192|--model ant --envs-per-gpu 8192|loop"; do
  IFS='|' read NAME ARGS FORM <<< "$C"
  i=0
  for CTRS in "$SQ1" "$SQ2" "$SQ3"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/sq_${NAME}_$i -o p -- python bench.py $ARGS --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events > $O/sq_${NAME}_$i.log 2>&1
  done
  if [ $FORM = loop ]; then python tools/pmc_loop_summary.py 200 $O/sq_${NAME}_* > $P/r04_${NAME}_f64_sq_counters.txt 2>&1
  else python tools/pmc_summary.py $O/sq_${NAME}_* > $P/r04_${NAME}_f64_sq_counters.txt 2>&1; fi
  grep -v '^# kernel' $P/r04_${NAME}_f64_sq_counters.txt | cut -c1-150
  rm -rf $O/sq_${NAME}_*/
done
# kernel trace + timeline of one rank through the exchange (the default form), at the driver's shape and in steady state
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_fg -o k -- python bench.py --no-cpu-baseline --no-secondary --steps 192 --warmup 64 --force-gather --spin-up-steps 0 > $O/kt_fg.log 2>&1
DB=$(ls $O/kt_fg/*.db $O/kt_fg/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > $P/r04_one_rank_exchange_kernel_stats.txt 2>&1
python tools/rocprof_timeline.py "$DB" 150 > $P/r04_one_rank_exchange_timeline.txt 2>&1
python tools/ring_overlap.py "$DB" > $P/r04_one_rank_exchange_overlap.txt 2>&1
rm -rf $O/kt_fg
head -12 $P/r04_one_rank_exchange_kernel_stats.txt | cut -c1-160
cat $P/r04_one_rank_exchange_overlap.txt | head -20 | cut -c1-200
# tds_hip::VectorizedEnv from C++ (the harness carries the class compiled against the reference's headers)
timeout 300 python - > $P/r04_cpp_vectorized_env_rates.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import reflib
for name, n, k in (("ant", 4096, 1000), ("ant", 4096, 20), ("ant", 1024, 1000), ("laikago", 4096, 200)):
    r = reflib.vecenv_hip_bench(name, n, k)
    print(f"tds_hip::VectorizedEnv<{name}> x{n}, {k} steps per device call, env-steps/s: " + "  ".join(f"{a}={b:.3e}" for a, b in r.items()))
PY
cat $P/r04_cpp_vectorized_env_rates.txt
timeout 120 tools/ubench/wait_value > $P/r04_ubench_wait_value.txt 2>&1
tail -12 $P/r04_ubench_wait_value.txt
