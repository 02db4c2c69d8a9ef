#!/bin/bash
# config 2 (pendulum5, float records / double arithmetic) by batch size: us per step — a lone wavefront per SIMD up to 4096
# environments (16 lanes each): the step is the latency of the 5-link chain, whatever the number of wavefronts
for N in ${SIZES:-1024 2048 4096 8192 16384 32768}; do
  timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 500 --warmup 50 --model pendulum5 --dtype f32 --envs-per-gpu $N > /tmp/p.json 2>/tmp/p.err
  python3 -c "
import json;d=json.load(open('/tmp/p.json'));print($N,'envs: value %.4g us/step %.2f'%(d['value'],1e3*d['ms_per_step']))" || tail -3 /tmp/p.err
done
