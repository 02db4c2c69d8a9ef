#!/bin/bash
# Runs ON THE GPU BOX: bench.py --gpus 8 with EIGHT ranks on the ONE GPU of the box (eight processes, real hipIpc mappings of
# each other's gathered rings; the communicator that carries the handles is tests/stub_rccl, because the real RCCL refuses
# several ranks on one device) — a dry run of the driver's N = 8 command: config 5 (8192 environments per rank) as `value`,
# the 4096-per-rank line beside it, seven peers per rank.  Not a measurement: eight launches time-share one GPU.
#     usage: tools/eight_rank_dry_run.sh [ranks]     ->  gpurun_out/profiles/<tag>_eight_rank_dry_run.txt  (TAG env, default rXX)
export TMPDIR=/tmp
N=${1:-8}
# (fewer ranks: TDS_BENCH_FORCE_CONFIG5=1 makes them walk the N = 8 flow — eight processes on one GPU spend most of their time in
#  the bounded waits of the exchange, one launch filling the GPU at a time: r05w gave up after 400 s)
[ "$N" != 8 ] && export TDS_BENCH_FORCE_CONFIG5=1
TAG=${TAG:-rXX}
O=gpurun_out/work_$TAG/eight_rank
P=gpurun_out/profiles
mkdir -p $O $P
g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/libstub_rccl.so tests/stub_rccl/stub_rccl.cpp -L/opt/rocm/lib -lamdhip64 -lrt -lpthread || exit 1
TDS_HIP_RCCL_LIB=/tmp/libstub_rccl.so TDS_BENCH_ONE_DEVICE=1 TDS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 ${TIMEOUT:-400} \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29551 \
  bench.py --gpus $N --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --spin-up-steps 0 > $O/line.json 2> $O/line.err
echo "rc=$?"
tail -4 $O/line.err | cut -c1-300
python3 - $O/line.json <<'PY' | tee $P/${TAG}_eight_rank_dry_run.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d["config"]
    print("bench.py --gpus %d, all ranks on ONE GPU (dry run of the driver's command, not a measurement)" % d["n_gpus"])
    print("  metric %s  value %.4g  n_gpus %d  steps %d  scaling %s  ms_per_step %.4f" % (d["metric"], d["value"], d["n_gpus"], d["steps"], d["scaling"], d["ms_per_step"]))
    print("  config.workload:", c["workload"][:160])
    print("  exchange_form:", c.get("exchange_form"), "| peers per rank:", c.get("peers"), "| parallelism:", c.get("parallelism", "")[:120])
    e = d.get("envs_4096_per_gpu")
    if e: print("  envs_4096_per_gpu:", {k: e[k] for k in e if k in ("value", "ms_per_step", "exchange_form", "error")})
    print("  finite:", d.get("finite"), "nonfinite_envs:", d.get("nonfinite_envs"))
except Exception as ex:
    print("NO LINE:", ex)
PY
