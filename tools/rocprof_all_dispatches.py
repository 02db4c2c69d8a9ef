#!/usr/bin/env python3
"""Every dispatch of a rocprofv3 (rocpd sqlite) kernel trace in start order: start, duration, queue, short name.
usage: python tools/rocprof_all_dispatches.py <results.db> [first=0] [count=200]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
st = "start" if "start" in cols else "start_timestamp"
en = "end" if "end" in cols else "end_timestamp"
q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else "0")
rows = db.execute(f"select name, {st}, {en}, {q}, grid_x from kernels order by {st}").fetchall()
print(f"# {len(rows)} dispatches")
if first < 0:
    first = max(0, len(rows) + first)
t0 = rows[first][1]
for name, s, e, qq, gx in rows[first:first + count]:
    short = re.sub(r"\(anonymous namespace\)::", "", name.split("(")[0])[-70:]
    m = re.search(r"tds_step_kernel<([^>]*)>", name)
    if m:
        short = "tds_step_kernel<" + m.group(1) + ">"
    print(f"{(s - t0) / 1e3:10.2f} us  dur {(e - s) / 1e3:9.2f}  q={qq}  grid={gx:7d}  {short}")
