#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc CSV output (one directory per pass) to per-dispatch means for the step kernel.
usage: python tools/pmc_summary.py <dir> [<dir> ...]   (each holds *_counter_collection.csv)"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(list)
grid = None
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float))
        for row in csv.DictReader(open(f)):
            if "tds_step_kernel" not in row["Kernel_Name"]:
                continue
            kname = row["Kernel_Name"]
            per[row["Dispatch_Id"]][row["Counter_Name"]] += float(row["Counter_Value"])
            grid = (row.get("Grid_Size"), row.get("Workgroup_Size"), row.get("LDS_Block_Size"))
        for disp in per.values():
            for k, v in disp.items():
                acc[k].append(v)
print(f"# mean per dispatch of tds_step_kernel (grid, wg, lds) = {grid}")
try:
    print(f"# kernel: {kname[:160]}")
except NameError:
    pass
waves = None
if "SQ_WAVES" in acc:
    waves = sum(acc["SQ_WAVES"]) / len(acc["SQ_WAVES"])
for k in sorted(acc):
    m = sum(acc[k]) / len(acc[k])
    extra = f"   per wave {m / waves:10.1f}" if waves else ""
    print(f"{k:28s} {m:14.1f}{extra}   ({len(acc[k])} dispatches)")
