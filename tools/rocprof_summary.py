#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result database as text: per-kernel call count, total /
average duration, and the launch geometry + register/LDS footprint of the step kernel.
usage: python tools/rocprof_summary.py <results.db> [> profiles/<name>.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>7}  kernel")
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{calls:7d} {tot:12.2f} {avg:10.3f} {pct:7.2f}  {name[:150]}")
row = db.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, "
                 "min(duration), avg(duration), max(duration), count(*) from kernels where (name like '%tds_step_kernel%' or name like '%tds_quad_kernel%' or name like '%tds_oct_kernel%' or name like '%tds_chain_kernel%') "
                 "group by name, grid_x").fetchall()
for r in row:
    # vgpr_count as rocprofv3 records it is the ALLOCATION in granules of this dispatch (wave64 on the unified 512-entry
    # file: half the per-lane register count), not the compiler's VGPR count: 120 <-> 240 allocated.  The compiler's
    # numbers (VGPR / AGPR / scratch / occupancy per instantiation) are in profiles/*kernel_resources*.txt
    # (tools/kernel_resources.sh, -Rpass-analysis=kernel-resource-usage).
    print(f"\n# {r[0][:150]}\n#   grid={r[1]} wg={r[2]} lds_bytes/wg={r[3]} scratch={r[4]} "
          f"rocprof_vgpr_field={r[5]} (x2 = allocated registers per lane) rocprof_agpr_field={r[6]} sgpr={r[7]}"
          f"\n#   duration ns: min={r[8]} avg={r[9]:.0f} max={r[10]} over {r[11]} dispatches")
