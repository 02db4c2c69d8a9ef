#!/bin/bash
# Runs ON THE GPU BOX: cost of the record exchange machinery on one GPU (device-copy stand-in for the all-gather)
# as a function of the number of steps whose records travel together.
for B in 8 16 32 64; do
  python bench.py --force-gather --gather-every $B --steps 960 --warmup 64 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gather-every', $B, '%.4e' % d['value'], round(d['ms_per_step']*1e3, 2), 'us/step')"
done
