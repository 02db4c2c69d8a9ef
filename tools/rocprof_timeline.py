#!/usr/bin/env python3
"""Timeline excerpt of a rocprofv3 (rocpd sqlite) kernel trace: dispatches in start order with start / duration / gap to
the previous dispatch of the same queue, for a window in the middle of the run.
usage: python tools/rocprof_timeline.py <results.db> [n_rows=160] [name_filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 160
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("# columns of `kernels`:", cols)
pick = lambda *names: next((c for c in names if c in cols), None)
c_start, c_end, c_dur = pick("start", "start_timestamp"), pick("end", "end_timestamp"), pick("duration")
c_q, c_s = pick("queue_id", "queue"), pick("stream_id", "stream")
sel = ", ".join(c for c in ("name", c_start, c_end, c_dur, c_q, c_s, "grid_x") if c)
rows = db.execute(f"select {sel} from kernels order by {c_start}").fetchall()
print(f"# {len(rows)} dispatches; showing {n} from the middle")
mid = max(0, len(rows) // 2 - n // 2)
t0 = rows[mid][1]
last_end = {}
for r in rows[mid:mid + n]:
    name, st, en = r[0], r[1], r[2]
    q = r[4] if len(r) > 4 else 0
    gap = (st - last_end[q]) if q in last_end else 0
    last_end[q] = en
    short = name.split("(")[0][-60:]
    print(f"{(st - t0) / 1e3:10.2f} us  dur {(en - st) / 1e3:8.2f} us  gap {gap / 1e3:8.2f}  q={q} s={r[5] if len(r) > 5 else ''} grid={r[-1]}  {short}")
