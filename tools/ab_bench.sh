#!/bin/bash
# Runs ON THE GPU BOX: A/B of library builds (tiny-differentiable-simulator_amd/libtds_hip_<tag>.so) on the bench.
# usage: tools/ab_bench.sh <tag> [<tag> ...]   -> one line per (tag, envs) in gpurun_out/ab.log
mkdir -p gpurun_out
for rep in 1 2; do
for t in "$@"; do
  for n in 4096 16384; do
    v=$(TDS_HIP_LIB=$PWD/tiny-differentiable-simulator_amd/libtds_hip_$t.so python bench.py --steps 400 --warmup 50 --envs-per-gpu $n --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4g %.2f us' % (d['value'], d['roofline']['kernel_ms_avg']*1e3))")
    echo "$t envs=$n rep=$rep: $v" | tee -a gpurun_out/ab.log
  done
done
done
