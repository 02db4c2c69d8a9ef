export TMPDIR=/tmp
O=gpurun_out/work_co; mkdir -p $O
timeout 250 rocprofv3 --kernel-trace -d $O/kt -o k -- python tools/call_overhead.py 20 > $O/log.txt 2>&1
DB=$(ls $O/kt/*.db $O/kt/*/*.db 2>/dev/null | head -1)
python3 - "$DB" <<'PY'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, grid_x from kernels order by start").fetchall()
# group: for each 20-step loop launch (dur 200..500us, LP=1 W2), print duration and the gap to previous kernel end / next
out=[]
for i,(n,a,b,g) in enumerate(rows):
    if "tds_step_kernel" in n and "false, 1, 0, true" in n and 150e3 < (b-a) < 600e3:
        prev = rows[i-1]; nxt = rows[i+1] if i+1 < len(rows) else None
        out.append(((b-a)/1e3, (a-prev[2])/1e3, prev[0].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:28], ((nxt[1]-b)/1e3 if nxt else -1), (nxt[0].replace("void ","").replace("(anonymous namespace)::","").split("(")[0][:28] if nxt else "")))
print("20-step loop launches in order: dur us | gap before (prev kernel) | gap after (next kernel)")
import itertools
for k,(d,gb,pn,ga,nn) in enumerate(out):
    if k % 9 == 0: print("%4d  dur %7.1f  before %8.1f (%s)  after %8.1f (%s)"%(k,d,gb,pn,ga,nn))
import statistics
n=len(out); print("count",n)
PY
