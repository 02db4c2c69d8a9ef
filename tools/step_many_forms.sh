#!/bin/bash
# Runs ON THE GPU BOX: env-steps/s of tds_hip_step_many by launch form — hipGraphs of single-step launches as 1, 2, 3
# environment chains, and ONE launch of the step-loop kernel — for the usual configs.
export TMPDIR=/tmp
CONFIGS=(
  "ant2048_f64|--model ant --envs-per-gpu 2048"
  "ant4096_f64|--model ant --envs-per-gpu 4096"
  "ant8192_f64|--model ant --envs-per-gpu 8192"
  "ant16384_f64|--model ant --envs-per-gpu 16384"
  "pendulum5_4096_f32rec|--model pendulum5 --envs-per-gpu 4096 --dtype f32"
  "laikago_soft8192_f64|--model laikago_soft --envs-per-gpu 8192"
)
fmt='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%-26s %-16s %.4g  %.2f us" % (sys.argv[1], sys.argv[2], d["value"], 1000*d["ms_per_step"]))'
echo "# bench.py --step-many-form graph --chains C | --step-many-form loop (1000 steps per call), env-steps/s and us per step"
for C in "${CONFIGS[@]}"; do
  NAME=${C%%|*}; ARGS=${C##*|}
  for CH in 1 2 3; do
    python bench.py $ARGS --step-many-form graph --chains $CH --no-cpu-baseline 2>/dev/null | python -c "$fmt" $NAME "graphs, chains $CH"
  done
  python bench.py $ARGS --step-many-form loop --no-cpu-baseline 2>/dev/null | python -c "$fmt" $NAME "one loop launch"
done
echo "# TDS_HIP_GRAM=1 (contact solve in Gram form on the f64 matrix cores), ant4096_f64, graphs with 2 / 1 chains"
for CH in 2 1; do
  TDS_HIP_GRAM=1 python bench.py --step-many-form graph --chains $CH --no-cpu-baseline 2>/dev/null | python -c "$fmt" "ant4096_f64 gram" "graphs, chains $CH"
done
