#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/ab2
P=$PWD/tiny-differentiable-simulator_amd
for rep in 1 2; do
for n in 4096 8192; do
  (cd ab_r1 && timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu $n 2>/dev/null | tail -1 > ../gpurun_out/ab2/r1_ant${n}_$rep.json)
  TDS_HIP_W2=0 timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu $n 2>/dev/null | tail -1 > gpurun_out/ab2/cur_ant${n}_$rep.json
  for v in "$@"; do
    TDS_HIP_LIB=$P/libtds_hip_$v.so TDS_HIP_W2=0 timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu $n 2>/dev/null | tail -1 > gpurun_out/ab2/${v}_ant${n}_$rep.json
  done
done
done
for f in gpurun_out/ab2/*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], '%.3f us'%(d['roofline']['kernel_ms_avg']*1e3))" 2>&1 | tail -1)"; done
