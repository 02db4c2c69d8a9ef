#!/bin/bash
# the 16-lane quadruped kernel: its tests (+ the ring tests of the Laikago models), config 4's bench lines
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-r05d}
mkdir -p $O
timeout 900 python -m pytest tests/test_quad.py tests/test_rings.py -q --timeout 600 -s -k "quad or laikago" 2>&1 | tail -40 > $O/pytest_quad.log
tail -25 $O/pytest_quad.log | cut -c1-300
for A in "" "--option step_many_loop=1" "--auto-reset" "--auto-reset --option step_many_loop=0" "--option quad=0"; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 1000 --warmup 100 --model laikago_soft --envs-per-gpu 8192 $A > $O/bench_laikago.json 2> $O/bench_laikago.err
  python3 - $O/bench_laikago.json "$A" <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("[%s] value %.4g us/step %.2f nonfinite %s kernel_ms %.4f"%(sys.argv[2],d["value"],1e3*d["ms_per_step"],d["nonfinite_envs"],d["roofline"]["kernel_ms_avg"]), d["config"]["launch"][:90])
except Exception as e:
    print("ERR",e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
done
