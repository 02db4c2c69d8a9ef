#!/bin/bash
# Runs ON THE GPU BOX: the auto-reset ring launches with the main wavefront at priority 0 / 3 (slots 1 / 2) and the library's own.
export TMPDIR=/tmp
O=gpurun_out/r04m
P=gpurun_out/profiles
mkdir -p $O $P
NS="timeout 120 python bench.py --no-cpu-baseline --no-secondary --auto-reset"
{
for K in "1000 100" "20 5"; do
  set -- $K
  for V in "library|" "slot1_prio0|--option alt_build=1" "slot2_prio3|--option alt_build=2" "one_wave_loop|--option loop_w2=0"; do
    IFS='|' read NAME ARGS <<< "$V"
    $NS --steps $1 --warmup $2 $ARGS > $O/ar_${NAME}_$1.json 2> $O/ar_${NAME}_$1.err
    echo "auto-reset, $1 steps, $NAME: $(python -c "import json;d=json.loads(open('$O/ar_${NAME}_$1.json').read().strip().splitlines()[-1]);print('%.4g env-steps/s %.2f us/step'%(d['value'],1000*d['ms_per_step']), d['config']['auto_reset'][-60:])" 2>&1 | tail -1)"
  done
done
} | tee $P/r04_auto_reset_priority.txt
