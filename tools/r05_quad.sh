#!/bin/bash
# the 16-lane quadruped kernel: its tests, then config 4's bench line with and without it
set -u
export TMPDIR=/tmp
O=gpurun_out/r05d
mkdir -p $O
timeout 900 python -m pytest tests/test_quad.py -q --timeout 600 -s 2>&1 | tail -40 > $O/pytest_quad.log
tail -25 $O/pytest_quad.log | cut -c1-300
for Q in 1; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 500 --warmup 50 --model laikago_soft --envs-per-gpu 8192 --option quad=$Q > $O/bench_laikago_quad$Q.json 2> $O/bench_laikago_quad$Q.err
  python3 - $O/bench_laikago_quad$Q.json $Q <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print("quad",sys.argv[2],"value %.4g us/step %.2f nonfinite %s"%(d["value"],1e3*d["ms_per_step"],d["nonfinite_envs"]), d["config"]["launch"][:80])
except Exception as e:
    print("ERR",e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
P
done
