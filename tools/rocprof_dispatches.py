#!/usr/bin/env python3
"""Every dispatch of the step kernels in a rocprofv3 (rocpd sqlite) kernel trace, in start order: which build, grid,
duration.  The step-loop launches of one bench.py run differ only in their step count (spin-up launches of a scratch
handle, warm-up, the timed launch, the secondary measurements): this listing is what lets the timed launch be told
apart from the others (its duration / its steps = the per-step figure of the bench line).
usage: python tools/rocprof_dispatches.py <results.db> [max_rows=80]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 80
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
st = "start" if "start" in cols else "start_timestamp"
en = "end" if "end" in cols else "end_timestamp"
rows = db.execute(f"select name, {st}, {en}, grid_x, workgroup_x from kernels where (name like '%tds_step_kernel%' or name like '%tds_quad_kernel%' or name like '%tds_oct_kernel%' or name like '%tds_chain_kernel%') order by {st}").fetchall()
print(f"# {len(rows)} dispatches of the step kernels (tds_step_kernel / tds_quad_kernel / tds_oct_kernel / tds_chain_kernel; showing the last {min(lim, len(rows))}); build = <T, TR, G, NDP, PROF, LP, KIND, W2>")
t0 = rows[0][1] if rows else 0
for name, s, e, gx, wx in rows[-lim:]:
    m = re.search(r"(?:tds_step_kernel|tds_quad_kernel|tds_oct_kernel|tds_chain_kernel)<([^>]*)>", name)
    print(f"{(s - t0) / 1e3:12.1f} us  dur {(e - s) / 1e3:10.2f} us  grid {gx:7d} wg {wx:4d}  <{m.group(1) if m else name[:60]}>")
