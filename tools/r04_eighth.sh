#!/bin/bash
# Runs ON THE GPU BOX (round 4, eighth call): slots on top of the new default (main wavefront at priority 3): priorities 0 / 1 /
# 2, two scheduler switches; what the timing harness of bench.py's 20-step region costs (events, NULL stream); then the whole
# GPU suite on this build — workers of their own (xdist) and a per-test limit, so that a launch that never ends costs one
# worker, not the call.
export TMPDIR=/tmp
O=gpurun_out/r04h
P=gpurun_out/profiles
mkdir -p $O $P
timeout 200 python tools/ab_slots.py --reps 5 > $P/r04_ab_slots2_ant4096.txt 2>&1; cat $P/r04_ab_slots2_ant4096.txt
B="timeout 120 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5"
{
for rep in 1 2; do
  for V in "default|" "no_events|--no-events" "own_stream|--stream own" "own_stream_no_events|--stream own --no-events"; do
    IFS='|' read NAME ARGS <<< "$V"
    $B $ARGS > $O/h_${NAME}_$rep.json 2> $O/h_${NAME}_$rep.err
    echo "$NAME $rep: $(python -c "
import json,sys
d=json.loads(open('$O/h_${NAME}_$rep.json').read().strip().splitlines()[-1]); print('%.4g env-steps/s  %.2f us/step  kernel_ms_avg %s'%(d['value'],1000*d['ms_per_step'],d['roofline'].get('kernel_ms_avg')))" 2>&1 | tail -1)"
  done
done
} | tee $P/r04_bench_20_step_harness_cost.txt
timeout 420 python -m pytest tests -m gpu -q -n 4 --timeout 90 --timeout-method=thread -p no:cacheprovider > $O/pytest_gpu.log 2>&1
tail -15 $O/pytest_gpu.log | cut -c1-300
cp $O/pytest_gpu.log $P/r04c_pytest_gpu.log
