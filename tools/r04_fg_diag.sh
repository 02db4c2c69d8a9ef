#!/bin/bash
# Runs ON THE GPU BOX: one rank through the ring exchange, 1024-step regions: with / without a communicator, with LDS left
# over for the exchange's kernels (na_cap = 5) — where the two-wavefront build with the constant table loses against itself.
export TMPDIR=/tmp
O=gpurun_out/r04k
P=gpurun_out/profiles
mkdir -p $O $P
NS="timeout 120 python bench.py --no-cpu-baseline --no-secondary --steps 1024 --warmup 128"
{
$NS > $O/plain.json 2> $O/plain.err; echo "plain N=1 launches: $(python -c "import json;d=json.loads(open('$O/plain.json').read().strip().splitlines()[-1]);print('%.2f us/step'%(1000*d['ms_per_step']))")"
for V in "rccl|" "rccl_na_cap5|--option na_cap=5" "rccl_one_wave|--option exchange_w2=0" "rccl_c64|--option shard_chunk=64"; do
  IFS='|' read NAME ARGS <<< "$V"
  $NS --force-gather $ARGS > $O/fg_$NAME.json 2> $O/fg_$NAME.err
  echo "$NAME: $(python -c "import json;d=json.loads(open('$O/fg_$NAME.json').read().strip().splitlines()[-1]);print('%.2f us/step'%(1000*d['ms_per_step']))" 2>&1 | tail -1)"
done
TDS_BENCH_RCCL_SINGLE=0 $NS --force-gather > $O/fg_nocomm.json 2> $O/fg_nocomm.err
echo "no communicator: $(python -c "import json;d=json.loads(open('$O/fg_nocomm.json').read().strip().splitlines()[-1]);print('%.2f us/step'%(1000*d['ms_per_step']))")"
TDS_BENCH_RCCL_SINGLE=0 $NS --force-gather --option na_cap=5 > $O/fg_nocomm5.json 2> $O/fg_nocomm5.err
echo "no communicator, na_cap=5: $(python -c "import json;d=json.loads(open('$O/fg_nocomm5.json').read().strip().splitlines()[-1]);print('%.2f us/step'%(1000*d['ms_per_step']))")"
} | tee $P/r04_one_rank_exchange_with_table.txt
