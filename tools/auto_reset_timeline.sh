#!/bin/bash
# auto_reset_when_done through the reset pool: the driver line's secondary key, the 1000-step rate under the pool options,
# and the dispatch timeline of a 1000-step run (where the refill passes sit between the chunks)
#     usage (on the GPU box): tools/auto_reset_timeline.sh <tag>     ->  gpurun_out/profiles/<tag>_auto_reset_*.txt
set -u
export TMPDIR=/tmp
TAG=${1:-rXX}
O=gpurun_out/work_$TAG/auto_reset
P=gpurun_out/profiles
mkdir -p $O $P
line() { python3 - "$1" "$2" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    a=d.get("auto_reset_rate")
    print("[%s] value %.4g  %.2f us/step  nonfinite %s"%(sys.argv[2],d["value"],1e3*d["ms_per_step"],d.get("nonfinite_envs")),
          ("auto_reset_rate %.4g = %.3f x value (calls %s, slowest %.4g fastest %.4g, done_in_last_step %s)"%(a["value"],a["value"]/d["value"],a.get("calls"),a.get("slowest_call",0),a.get("fastest_call",0),a.get("done_in_last_step")) if a and "value" in a else ""))
except Exception as e:
    print("[%s] ERR"%sys.argv[2],e); print(open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
}
{
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/driver.json 2> $O/driver.err; line $O/driver.json "driver line"
timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-secondary > $O/plain.json 2> $O/plain.err; line $O/plain.json "plain 1000"
for A in "" "--option pool_settle_loop=0" "--option pool_chunk=128" "--option pool_chunk=128 --option pool_settle_loop=0"; do
  timeout 300 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-secondary --auto-reset $A > $O/ar.json 2> $O/ar.err; line $O/ar.json "auto-reset 1000 $A"
done
} 2>&1 | tee $P/${TAG}_auto_reset_lines.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-secondary --auto-reset > $O/kt.log 2>&1
DB=$(ls $O/kt/*.db $O/kt/*/*.db 2>/dev/null | head -1)
python3 - "$DB" <<'PY' | tee $P/${TAG}_auto_reset_timeline.txt
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
st = "start" if "start" in cols else "start_timestamp"
en = "end" if "end" in cols else "end_timestamp"
rows = con.execute(f"select name, {st}, {en}, grid_x from kernels order by {st}").fetchall() if "grid_x" in cols else \
       [r + (0,) for r in con.execute(f"select name, {st}, {en} from kernels order by {st}").fetchall()]
# the last 1000-step call: from the 8th-last step-loop launch (LP = 1 build, > 1 ms) to the end
STEP = ("tds_step_kernel", "tds_oct_kernel", "tds_quad_kernel", "tds_chain_kernel")
# (the chunks: step-loop launches of 128 steps — longer than 0.5 ms whichever kernel runs them)
big = [i for i, r in enumerate(rows) if any(k in r[0] for k in STEP) and (r[2] - r[1]) > 5e5]
i0 = big[-8] if len(big) >= 8 else 0
t0 = rows[i0][1]
print("# bench.py --auto-reset --steps 1000: every dispatch of the timed call (chunks of 128 steps; between them the refill pass)")
print("# start us | duration us | grid | kernel")
prev_end = t0
agg = {}
for n, a, b, g in rows[i0:]:
    k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    print("%10.1f  %9.1f  %8d  %s" % ((a - t0) / 1e3, (b - a) / 1e3, g, k))
    agg[k.split("<")[0]] = agg.get(k.split("<")[0], 0) + (b - a) / 1e3
print("# totals by kernel (us):", {k: round(v, 1) for k, v in agg.items()})
print("# wall of the call: %.1f us" % ((rows[-1][2] - t0) / 1e3))
PY
