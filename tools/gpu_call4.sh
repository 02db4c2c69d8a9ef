#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/c4
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/c4/pytest_gpu.log
tail -8 gpurun_out/c4/pytest_gpu.log
timeout 600 python tools/auto_reset_modes.py > gpurun_out/c4/auto_reset_modes.txt 2>&1
timeout 600 python tools/rollout_modes.py > gpurun_out/c4/rollout_modes.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/c4/bench_ant4096.json 2> gpurun_out/c4/bench.err
timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 8192 > gpurun_out/c4/bench_ant8192.json 2>> gpurun_out/c4/bench.err
timeout 300 python bench.py --no-cpu-baseline --envs-per-gpu 16384 > gpurun_out/c4/bench_ant16384.json 2>> gpurun_out/c4/bench.err
timeout 300 python bench.py --no-cpu-baseline --model laikago_soft --envs-per-gpu 8192 > gpurun_out/c4/bench_laikago_soft8192.json 2>> gpurun_out/c4/bench.err
timeout 300 python bench.py --no-cpu-baseline --model humanoid --rollout-steps 20 > gpurun_out/c4/bench_humanoid4096.json 2>> gpurun_out/c4/bench.err
timeout 300 python bench.py --no-cpu-baseline --rollout-steps 100 > gpurun_out/c4/bench_ant4096_rollout.json 2>> gpurun_out/c4/bench.err
python tools/profile_phases.py ant 4096 > gpurun_out/c4/phases_ant4096.txt 2>&1
python tools/profile_phases.py ant 8192 > gpurun_out/c4/phases_ant8192.txt 2>&1
echo "--- auto reset"; cat gpurun_out/c4/auto_reset_modes.txt
echo "--- rollout"; cat gpurun_out/c4/rollout_modes.txt
for f in gpurun_out/c4/bench_*.json; do echo "$f: $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('%.4g'%d['value'], d['ms_per_step'], (d.get('on_device_rollout') or {}).get('value'))" 2>&1 | tail -1)"; done
cat gpurun_out/c4/phases_ant4096.txt
