#!/bin/bash
# Runs ON THE GPU BOX: A/B of library builds on ONE box, interleaved: tools/ab_libs.sh "<bench.py arguments>" <lib.so> [<lib.so> ...]
# (paths relative to tiny-differentiable-simulator_amd/; two rounds, one line per library and round)
ARGS=$1; shift
for rep in 1 2; do
  for L in "$@"; do
    v=$(TDS_HIP_LIB=$PWD/tiny-differentiable-simulator_amd/$L timeout 300 python bench.py --no-cpu-baseline --no-secondary $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4g env-steps/s %.2f us/step' % (d['value'], 1e3*d['ms_per_step']))")
    echo "$L [$ARGS] rep $rep: $v"
  done
done
