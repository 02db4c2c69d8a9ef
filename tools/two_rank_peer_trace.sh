#!/bin/bash
# Runs ON THE GPU BOX: bench.py --gpus 2 with TWO ranks on the ONE GPU of the box (two processes, real hipIpc mappings of each
# other's gathered rings; the communicator that carries the handles is tests/stub_rccl, because the real RCCL refuses two
# ranks on one device) under rocprofv3 --kernel-trace: which kernels an N = 2 run of the peer-store exchange consists of.
#     usage: tools/two_rank_peer_trace.sh <tag>      ->  gpurun_out/profiles/<tag>_two_rank_peer_*.txt
# What the trace is evidence for: per launch of the step-loop kernel one credit kernel in front and one arrival kernel behind
# it, per rank — and NO kernel of an exchange (all-gather, copy, wait) anywhere: the records cross between the ranks inside
# the step kernel.  Not a measurement of anything: two 1024-workgroup launches time-share one GPU.
export TMPDIR=/tmp
TAG=${1:-rXX}
O=gpurun_out/work_$TAG/two_rank
P=gpurun_out/profiles
mkdir -p $O $P
g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/libstub_rccl.so tests/stub_rccl/stub_rccl.cpp -L/opt/rocm/lib -lamdhip64 -lrt -lpthread || exit 1
# (one rocprofv3 per RANK, each with its own output directory: two processes writing one results database abort in SQLite
#  and rocprofv3 then waits for them for ever — r05g lost ten GPU-minutes to that; everything under a hard timeout)
cat > /tmp/two_rank_one.sh <<EOF
#!/bin/bash
exec rocprofv3 --kernel-trace -d $O/kt_\$RANK -o k -- python bench.py --gpus 2 --steps ${STEPS:-40} --warmup ${WARMUP:-10} --envs-per-gpu ${ENVS:-1024} --no-cpu-baseline --spin-up-steps 0
EOF
chmod +x /tmp/two_rank_one.sh
TDS_HIP_RCCL_LIB=/tmp/libstub_rccl.so TDS_BENCH_ONE_DEVICE=1 TDS_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 150 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29549 --no-python \
  /tmp/two_rank_one.sh > $O/line.json 2> $O/line.err
echo "rc=$?"; tail -3 $O/line.err | cut -c1-300
python - $O $TAG <<'PY' | tee $P/${2:-$TAG}_two_rank_peer_summary.txt
import glob, json, sqlite3, sys, collections
O, tag = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(O + "/line.json").read().strip().splitlines()[-1])
    print("bench.py --gpus 2, two processes on ONE GPU (dry run of the N > 1 flow, not a measurement):")
    print("   config.exchange_form =", d["config"].get("exchange_form"), "| peers per rank =", d["config"].get("peers"))
    print("   config.parallelism =", d["config"].get("parallelism"))
    print("   value = %.4g (two ranks time-share one GPU)" % d["value"])
except Exception as e:
    print("NO LINE:", e)
for db in sorted(glob.glob(O + "/kt_*/**/*.db", recursive=True)):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    kt = next((t for t in tabs if t == "kernels"), None)
    if not kt:
        continue
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    st = "start" if "start" in cols else "start_timestamp"
    en = "end" if "end" in cols else "end_timestamp"
    rows = con.execute(f"select name, {st}, {en} from kernels order by {st}").fetchall()
    if not rows:
        continue
    c = collections.Counter()
    t = collections.Counter()
    for n, a, b in rows:
        k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][-48:]
        c[k] += 1
        t[k] += (b - a)
    print("\n%s: %d dispatches" % ("/".join(db.split("/")[-2:]), len(rows)))
    for k, v in c.most_common(12):
        print("   %-50s x%-6d total %10.1f us" % (k, v, t[k] / 1e3))
    names = " ".join(c)
    print("   any RCCL / all-gather / copy kernel in this process:", any(s in names.lower() for s in ("nccl", "rccl", "allgather", "all_gather")))
    # the last step-loop launch and what surrounds it
    idx = [i for i, r in enumerate(rows) if "tds_step_kernel" in r[0]]
    if idx:
        i = max(idx, key=lambda j: rows[j][2] - rows[j][1])  # the longest one = the step-loop launch of the timed steps
        t0 = rows[i][1]
        print("   around the step-loop launch of the timed steps (the longest tds_step_kernel dispatch):")
        for n, a, b in rows[max(0, i - 3):i + 4]:
            print("     %+10.1f us  dur %9.1f us  %s" % ((a - t0) / 1e3, (b - a) / 1e3, n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]))
PY
