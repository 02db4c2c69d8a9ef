#!/bin/bash
# Runs ON THE GPU BOX (round 4, fourth call): GPU suite (per-slot progress counters); the step kernels compiled without
# MachineLICM (scratch-free step-loop builds) against the ordinary build, every benchmark config; what a launch of the ring
# exchange costs by chunk length and by where the helper wavefront signals.
export TMPDIR=/tmp
O=gpurun_out/r04d
mkdir -p $O gpurun_out/profiles
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
B="timeout 300 python bench.py --no-cpu-baseline --no-secondary"
V=$PWD/tiny-differentiable-simulator_amd/libtds_hip_xnolicm.so
for rep in 1 2; do
  for L in base nolicm; do
    [ $L = base ] && unset TDS_HIP_LIB || export TDS_HIP_LIB=$V
    $B --steps 1000 --warmup 100 > $O/ab_${L}_ant4096_1000_$rep.json 2> $O/ab_${L}_ant4096_1000_$rep.err
    $B --steps 20 --warmup 5 > $O/ab_${L}_ant4096_20_$rep.json 2> $O/ab_${L}_ant4096_20_$rep.err
    $B --steps 500 --warmup 50 --envs-per-gpu 8192 > $O/ab_${L}_ant8192_$rep.json 2> $O/ab_${L}_ant8192_$rep.err
    $B --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 > $O/ab_${L}_laikago_$rep.json 2> $O/ab_${L}_laikago_$rep.err
    $B --steps 500 --warmup 50 --model pendulum5 --dtype f64 > $O/ab_${L}_pendulum5_$rep.json 2> $O/ab_${L}_pendulum5_$rep.err
    $B --steps 1000 --warmup 100 --option loop_w2=0 > $O/ab_${L}_ant4096_onewave_$rep.json 2> $O/ab_${L}_ant4096_onewave_$rep.err
    $B --steps 500 --warmup 50 --envs-per-gpu 16384 > $O/ab_${L}_ant16384_$rep.json 2> $O/ab_${L}_ant16384_$rep.err
  done
done
unset TDS_HIP_LIB
FG="$B --steps 1024 --warmup 128 --force-gather"
for L in base nolicm; do
  [ $L = base ] && unset TDS_HIP_LIB || export TDS_HIP_LIB=$V
  for C in 64 256 1024; do
    TDS_BENCH_RCCL_SINGLE=0 $FG --option shard_chunk=$C > $O/fg_${L}_nocomm_c$C.json 2> $O/fg_${L}_nocomm_c$C.err
  done
  $FG --option shard_chunk=256 > $O/fg_${L}_rccl_c256.json 2> $O/fg_${L}_rccl_c256.err
  $FG > $O/fg_${L}_rccl_c64.json 2> $O/fg_${L}_rccl_c64.err
  TDS_BENCH_RCCL_SINGLE=0 $FG --option shard_chunk=256 --option exchange_w2=1 > $O/fg_${L}_nocomm_c256_w2.json 2> $O/fg_${L}_nocomm_c256_w2.err
  TDS_BENCH_RCCL_SINGLE=0 $FG --option shard_chunk=256 --option exchange_w2=1 --option ring_signal_late=1 > $O/fg_${L}_nocomm_c256_w2_late.json 2> $O/fg_${L}_nocomm_c256_w2_late.err
done
unset TDS_HIP_LIB
for f in $O/*.json; do echo "$(basename $f): $(python - "$f" <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('%.4g'%d['value'], '%.2f us/step'%(1000*d['ms_per_step']), 'nonfinite=%s'%d.get('nonfinite_envs'))
except Exception as e:
    print('ERR', e, open(sys.argv[1].replace('.json','.err')).read()[-300:])
P
)"; done
