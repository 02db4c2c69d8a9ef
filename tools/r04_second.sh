#!/bin/bash
# Runs ON THE GPU BOX (round 4, second call): the whole GPU suite on the new build (options API, in-place exchange, padded
# y rings, tds_hip::VectorizedEnv), then the measurements behind DESIGN section 7: one rank through the exchange in its
# forms, HBM write traffic with packed / line-padded y records, the C++ class's rates, what hipStreamWaitValue64 runs as.
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O gpurun_out/profiles
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > $O/bench_default20.json 2> $O/bench_default20.err
$B --steps 1000 --warmup 100 > $O/bench_1000.json 2> $O/bench_1000.err
$B --steps 1000 --warmup 100 --y-stride packed --no-secondary > $O/bench_1000_packed.json 2> $O/bench_1000_packed.err
FG="$B --steps 1000 --warmup 100 --force-gather --no-secondary"
$FG > $O/bench_fg1000.json 2> $O/bench_fg1000.err
$FG --option shard_wait=1 > $O/bench_fg1000_waitvalue.json 2> $O/bench_fg1000_waitvalue.err
$FG --option shard_inplace=0 > $O/bench_fg1000_sendring.json 2> $O/bench_fg1000_sendring.err
$FG --option exchange_w2=1 > $O/bench_fg1000_w2.json 2> $O/bench_fg1000_w2.err
$FG --option exchange_w2=1 --option shard_wait=1 > $O/bench_fg1000_w2_waitvalue.json 2> $O/bench_fg1000_w2_waitvalue.err
TDS_BENCH_RCCL_SINGLE=0 $FG > $O/bench_fg1000_nocomm.json 2> $O/bench_fg1000_nocomm.err
TDS_BENCH_RCCL_SINGLE=0 $FG --option exchange_w2=1 > $O/bench_fg1000_nocomm_w2.json 2> $O/bench_fg1000_nocomm_w2.err
$FG --envs-per-gpu 8192 --steps 500 > $O/bench_fg500_8192.json 2> $O/bench_fg500_8192.err
$B --steps 500 --warmup 50 --envs-per-gpu 8192 --no-secondary > $O/bench_8192.json 2> $O/bench_8192.err
$B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 --no-secondary > $O/bench_laikago.json 2> $O/bench_laikago.err
# HBM traffic of the driver's command and of the 1000-step region, packed / padded y records
for YS in line packed; do
  for K in 20 1000; do
    i=0
    for CTRS in FETCH_SIZE WRITE_SIZE; do
      i=$((i+1))
      W=$( [ $K = 20 ] && echo 5 || echo 100 )
      timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $O/pmc_${YS}_${K}_$i -o p -- python bench.py --steps $K --warmup $W --no-cpu-baseline --no-secondary --spin-up-steps 0 --no-events --y-stride $YS > $O/pmc_${YS}_${K}_$i.log 2>&1
    done
    python tools/pmc_loop_summary.py $K $O/pmc_${YS}_${K}_* > gpurun_out/profiles/r04_ant4096_f64_${K}_ystride_${YS}_pmc_traffic.txt 2>&1
    rm -rf $O/pmc_${YS}_${K}_*/
  done
done
grep -h -v '^# kernel' gpurun_out/profiles/r04_ant4096_f64_*_pmc_traffic.txt | cut -c1-140
# tds_hip::VectorizedEnv from C++ (the harness carries the class compiled against the reference's headers)
timeout 300 python - > gpurun_out/profiles/r04_cpp_vectorized_env_rates.txt 2>&1 <<'P'
import sys, os
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import reflib
for name, n, k in (("ant", 4096, 1000), ("ant", 1024, 1000), ("laikago", 4096, 200)):
    r = reflib.vecenv_hip_bench(name, n, k)
    print(f"tds_hip::VectorizedEnv<{name}> x{n}, {k} steps per device call, env-steps/s: " + "  ".join(f"{a}={b:.3e}" for a, b in r.items()))
P
cat gpurun_out/profiles/r04_cpp_vectorized_env_rates.txt
# what hipStreamWaitValue64 runs as on this ROCm: kernel trace of the probe
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_wait -o k -- tools/ubench/wait_value > $O/kt_wait.log 2>&1
DB=$(ls $O/kt_wait/*.db $O/kt_wait/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > gpurun_out/profiles/r04_ubench_wait_value_kernel_trace.txt 2>&1
rm -rf $O/kt_wait
head -12 gpurun_out/profiles/r04_ubench_wait_value_kernel_trace.txt | cut -c1-150
# kernel trace of one rank through the exchange (the default form)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_fg -o k -- python bench.py --no-cpu-baseline --no-secondary --steps 192 --warmup 64 --force-gather --spin-up-steps 0 > $O/kt_fg.log 2>&1
DB=$(ls $O/kt_fg/*.db $O/kt_fg/*/*.db 2>/dev/null | head -1)
python tools/rocprof_summary.py "$DB" > gpurun_out/profiles/r04_one_rank_exchange_kernel_stats.txt 2>&1
python tools/rocprof_timeline.py "$DB" 150 > gpurun_out/profiles/r04_one_rank_exchange_timeline.txt 2>&1
rm -rf $O/kt_fg
head -14 gpurun_out/profiles/r04_one_rank_exchange_kernel_stats.txt | cut -c1-150
for f in $O/bench_*.json; do echo "$(basename $f): $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    x=['%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step'])]
    for k in ('substep_fused','one_rank_with_exchange','auto_reset_rate'):
        if k in d and d[k]: x.append(k+'='+('%.4g'%d[k]['value'] if 'value' in d[k] else d[k].get('error','?')[:80]))
    x.append(str(d['config'].get('exchange_form')))
    print(' '.join(x))
except Exception as e:
    print('ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
P
)"; done
