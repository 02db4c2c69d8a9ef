#!/bin/bash
# Runs ON THE GPU BOX: bench.py's N > 1 flow — rendezvous, shards, ring exchange, the exchange-build selection, the timed
# region, the JSON line — with TWO ranks on the ONE GPU of the box through tests/stub_rccl (an all-gather over shared memory
# behind RCCL's symbol names; the real RCCL refuses two ranks on one device).  A dry run of the code path, never a number.
export TMPDIR=/tmp
O=gpurun_out/two_rank_stub
mkdir -p $O gpurun_out/profiles
g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/libstub_rccl.so tests/stub_rccl/stub_rccl.cpp -L/opt/rocm/lib -lamdhip64 -lrt -lpthread || exit 1
TDS_HIP_RCCL_LIB=/tmp/libstub_rccl.so TDS_BENCH_ONE_DEVICE=1 TDS_BENCH_BACKEND=gloo timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29547 bench.py --gpus 2 --steps ${STEPS:-20} --warmup ${WARMUP:-5} --envs-per-gpu ${ENVS:-1024} --no-cpu-baseline > $O/line.json 2> $O/line.err
echo "rc=$?"; tail -4 $O/line.err | cut -c1-300
python - <<'PY' | tee gpurun_out/profiles/r04_two_rank_bench_dry_run.txt
import json
try:
    d = json.loads(open("gpurun_out/two_rank_stub/line.json").read().strip().splitlines()[-1])
    print("two ranks on one GPU through the stub all-gather (dry run of bench.py --gpus 2, not a measurement):")
    for k in ("metric", "n_gpus", "steps", "warmup", "scaling"):
        print("  ", k, "=", d[k])
    print("   value = %.4g (two ranks time-share one GPU and the stub's all-gather runs through host memory)" % d["value"])
    print("   config.parallelism =", d["config"].get("parallelism"))
    print("   config.exchange_form =", d["config"].get("exchange_form"))
    print("   config.exchange_tune =", d["config"].get("exchange_tune"))
except Exception as e:
    print("NO LINE:", e)
PY
