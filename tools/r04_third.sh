#!/bin/bash
# Runs ON THE GPU BOX (round 4, third call): the whole GPU suite; how the exchange's kernels overlap the step-loop launch
# (one-wave / two-wavefront build under the exchange); same-box A/B of round 3's library, this round's and the s_setprio
# experiment at the headline size.
export TMPDIR=/tmp
O=gpurun_out/r04c
mkdir -p $O gpurun_out/profiles
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
for V in w1 w2 w2_nocomm; do
  case $V in w1) A="1 4096 4";; w2) A="1 4096 4 exchange_w2=1";; w2_nocomm) A="0 4096 4 exchange_w2=1";; esac
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$V -o k -- python tools/trace_ring_exchange.py $A > $O/kt_$V.log 2>&1
  DB=$(ls $O/kt_$V/*.db $O/kt_$V/*/*.db 2>/dev/null | head -1)
  { echo "# tools/trace_ring_exchange.py $A"; grep 'us per step' $O/kt_$V.log; python tools/ring_overlap.py "$DB"; python tools/rocprof_summary.py "$DB" | head -8; } > gpurun_out/profiles/r04_ring_exchange_overlap_$V.txt 2>&1
  rm -rf $O/kt_$V
  cut -c1-230 gpurun_out/profiles/r04_ring_exchange_overlap_$V.txt
done
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
for rep in 1 2; do
  $B --steps 1000 --warmup 100 > $O/bench_new_1000_$rep.json 2> $O/bench_new_1000_$rep.err
  $B --steps 20 --warmup 5 > $O/bench_new_20_$rep.json 2> $O/bench_new_20_$rep.err
  (cd ab_r03 && $B --steps 1000 --warmup 100) > $O/bench_r03_1000_$rep.json 2> $O/bench_r03_1000_$rep.err
  (cd ab_r03 && $B --steps 20 --warmup 5) > $O/bench_r03_20_$rep.json 2> $O/bench_r03_20_$rep.err
  TDS_HIP_LIB=$PWD/tiny-differentiable-simulator_amd/libtds_hip_xprio3.so $B --steps 1000 --warmup 100 > $O/bench_prio3_1000_$rep.json 2> $O/bench_prio3_1000_$rep.err
  TDS_HIP_LIB=$PWD/tiny-differentiable-simulator_amd/libtds_hip_xprio3.so $B --steps 20 --warmup 5 > $O/bench_prio3_20_$rep.json 2> $O/bench_prio3_20_$rep.err
done
$B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/bench_new_8192.json 2> $O/bench_new_8192.err
(cd ab_r03 && $B --steps 500 --warmup 50 --envs-per-gpu 8192) > $O/bench_r03_8192.json 2> $O/bench_r03_8192.err
for f in $O/bench_*.json; do echo "$(basename $f): $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'kernel_ms_avg=%.4f'%d['roofline']['kernel_ms_avg'])
except Exception as e:
    print('ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
P
)"; done
