#!/usr/bin/env python3
"""rocprofv3 --pmc CSV output of a run whose timed region is ONE step-loop launch (tds_hip_step_many of K steps): the
counters of the longest step-kernel dispatch (tds_step_kernel, tds_quad_kernel, tds_oct_kernel or tds_chain_kernel), and per step.
K = 0: the run's timed region is single-step launches (one per step) — the MEAN over the dispatches of the kernel that ran
most often, per launch = per step.
usage: python tools/pmc_loop_summary.py K <dir> [<dir> ...]"""
import csv
import glob
import os
import sys

K = int(sys.argv[1])
best = {}
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if not any(k in row["Kernel_Name"] for k in ("tds_step_kernel", "tds_quad_kernel", "tds_oct_kernel", "tds_chain_kernel")):
                continue
            key = row["Dispatch_Id"]
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            e = per.setdefault(key, {"dur": dur, "name": row["Kernel_Name"], "ctr": {}})
            e["ctr"][row["Counter_Name"]] = e["ctr"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        if per and K == 0:
            names = [e["name"] for e in per.values()]
            name = max(set(names), key=names.count)
            es = [e for e in per.values() if e["name"] == name]
            for k in es[0]["ctr"]:
                vals = [e["ctr"][k] for e in es if k in e["ctr"]]
                best[k] = (sum(vals) / len(vals), sum(e["dur"] for e in es) / len(es), name, len(vals))
        elif per:
            top = max(per.values(), key=lambda e: e["dur"])
            for k, v in top["ctr"].items():
                best[k] = (v, top["dur"], top["name"])
if K == 0:
    print("# mean over the single-step launches of the kernel that ran most often (one launch = one step)")
    for k in sorted(best):
        v, dur, name, cnt = best[k]
        print(f"# kernel: {name[:150]}")
        print(f"{k:28s} {v:14.1f} per launch ({dur / 1000.0:.1f} us, mean of {cnt} launches)   {v:12.3f} per step")
    sys.exit(0)
print(f"# longest step-kernel dispatch of each pass = the step-loop launch of the {K} timed steps")
for k in sorted(best):
    v, dur, name = best[k]
    print(f"# kernel: {name[:150]}")
    print(f"{k:28s} {v:14.1f} per launch ({dur / 1000.0:.1f} us)   {v / K:12.3f} per step")
