#!/usr/bin/env python3
"""Does the exchange run BESIDE the step-loop launch?  From a rocprofv3 (rocpd sqlite) kernel trace of
tools/trace_ring_exchange.py: for every step-loop launch (the tds_step_kernel dispatches of > 100 us) the kernels of the
communication stream (wait kernels, RCCL's / the runtime's copy and fill kernels) that STARTED inside its interval, and
where in the interval (fraction of the launch's duration) the first, the median and the last of them started.
usage: python tools/ring_overlap.py <results.db>"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
st = "start" if "start" in cols else "start_timestamp"
en = "end" if "end" in cols else "end_timestamp"
rows = db.execute(f"select name, {st}, {en}, grid_x from kernels order by {st}").fetchall()
steps = [(s, e, n) for n, s, e, g in rows if "tds_step_kernel" in n and e - s > 100_000]
others = [(s, e, n.split("(")[0][-40:]) for n, s, e, g in rows if "tds_step_kernel" not in n]
print(f"# {len(rows)} dispatches, {len(steps)} step-loop launches of > 100 us")
for s, e, n in steps:
    inside = [(os_ - s) / (e - s) for os_, oe, on in others if s <= os_ < e]
    durs = [(oe - os_) / 1e3 for os_, oe, on in others if s <= os_ < e]
    kind = n[n.find("tds_step_kernel<"):][:70]
    if inside:
        q = np.quantile(inside, [0.0, 0.5, 1.0])
        print(f"launch {(e - s) / 1e3:8.1f} us  {kind}: {len(inside):4d} other kernels started inside it, at "
              f"{q[0]:.2f} / {q[1]:.2f} / {q[2]:.2f} of its duration (first / median / last); their durations "
              f"median {np.median(durs):.1f} us, max {np.max(durs):.1f} us")
    else:
        print(f"launch {(e - s) / 1e3:8.1f} us  {kind}: no other kernel started inside it")
