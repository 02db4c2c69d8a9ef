#!/bin/bash
# Runs ON THE GPU BOX (round 4, ninth call): config 4's kernel compiled for three wavefronts per SIMD (168 VGPR + 440 B of
# scratch) against the two-per-SIMD build; bench.py's 20-step region with a keep-warm launch in front of it.
export TMPDIR=/tmp
O=gpurun_out/r04i
P=gpurun_out/profiles
mkdir -p $O $P
timeout 200 python tools/ab_slots.py --model laikago_soft --envs 8192 --slots 0,1,2 --steps 200 --short 20 --reps 3 > $P/r04_ab_slots_laikago8192_waves3.txt 2>&1; cat $P/r04_ab_slots_laikago8192_waves3.txt
B="timeout 120 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5"
for rep in 1 2 3; do
  $B > $O/warm_$rep.json 2> $O/warm_$rep.err
  python -c "
import json
d=json.loads(open('$O/warm_$rep.json').read().strip().splitlines()[-1]); print('keep-warm launch, run $rep: %.4g env-steps/s  %.2f us/step  kernel_ms_avg %s'%(d['value'],1000*d['ms_per_step'],d['roofline'].get('kernel_ms_avg')))"
done | tee -a $P/r04_bench_20_step_harness_cost.txt
