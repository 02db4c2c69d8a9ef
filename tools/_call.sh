#!/bin/bash
export TMPDIR=/tmp
for sp in 0 2000 2000 0 8000; do
echo "spin-up $sp: $(timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --spin-up-steps $sp 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g %.2f us/step; events %.1f us'%(d['value'], 1000*d['ms_per_step'], 1000*d['roofline']['kernel_ms_avg']))")"
done
