#!/bin/bash
export TMPDIR=/tmp
# N > 1 code path of bench.py on ONE device (debugging aid, never for numbers): RCCL refuses two ranks on one GPU, so
# this exercises the torch.distributed fallback
TDS_BENCH_ONE_DEVICE=1 TDS_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/fallback2.json 2> gpurun_out/fallback2.err
tail -5 gpurun_out/fallback2.err; cat gpurun_out/fallback2.json | cut -c1-1500
