#!/usr/bin/env python3
"""profiles/<tag>_*_pmc_traffic.txt and *_sq_counters*.txt (tools/collect.sh: sections traffic / sq / others)  ->  the two
tables bench.py reads: profiles/pmc_traffic.json (HBM bytes per launch) and profiles/sq_counters.json (VALU issue fraction).
usage: python tools/profiles_to_json.py <tag>        (after copying gpurun_out/profiles/<tag>_* into profiles/)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1]
SIMDS, PEAK_MHZ = 1024, 2400.0  # MI355X: 256 CUs x 4 SIMDs, peak engine clock (MI355X_MICROARCH.md)


def read(path):
    """{counter: (per_launch, us, per_step)}, kernel name"""
    out, kern = {}, None
    for line in open(path):
        if line.startswith("# kernel:"):
            kern = re.search(r"(tds_\w+_kernel<[^>]*>)", line)
            kern = kern.group(1) if kern else line[10:80].strip()
            continue
        mm = re.match(r"(\w+)\s+([\d.]+) per launch \(([\d.]+) us[^)]*\)\s+([\d.]+) per step", line)
        if mm:
            out[mm.group(1)] = (float(mm.group(2)), float(mm.group(3)), float(mm.group(4)))
    return out, kern


def rel(path):
    return "profiles/" + os.path.basename(path)


traffic = json.load(open(os.path.join(P, "pmc_traffic.json")))
sq_path = os.path.join(P, "sq_counters.json")
sq = json.load(open(sq_path)) if os.path.exists(sq_path) else {}
sq["_comment"] = ("per launch of the dominant kernel, from separate rocprofv3 --pmc passes (tools/collect.sh sections sq / others; "
                  "tools/profiles_to_json.py): valu_issue_frac = SQ_INSTS_VALU x 4 cycles (a 64-lane VALU instruction occupies its "
                  "16-lane SIMD for 4 cycles, f64 FMA included) / (%d SIMDs x launch duration x %.0f MHz peak clock): the share of "
                  "the GPU's VALU issue slots the launch fills; wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
                  "valu_insts_per_env_step = SQ_INSTS_VALU / (environments x steps)" % (SIMDS, PEAK_MHZ))

# (file stem after the tag, model, n, dtype key of the json, steps of the launch (0 = single-step launches), slot in pmc_traffic.json)
CASES = [
    ("ant4096_f64_20", "ant", 4096, "f64", 20, "rings_short"), ("ant4096_f64_1000", "ant", 4096, "f64", 1000, "rings_long"),
    ("ant8192_f64", "ant", 8192, "f64", 500, "rings_both"),
    ("laikago_soft8192_f64", "laikago_soft", 8192, "f64", 0, "single"),  # (tags up to r06d)
    ("laikago_soft8192_f64_single", "laikago_soft", 8192, "f64", 0, "single"),
    ("laikago_soft8192_f64_loop", "laikago_soft", 8192, "f64", 500, "rings_both"),
    ("laikago_soft4096_f64_loop", "laikago_soft", 4096, "f64", 500, "rings_both"),
    ("pendulum5_4096_f32rec", "pendulum5", 4096, "f32", 500, "rings_both"),
]
for stem, model, n, dt, K, slot in CASES:
    f = os.path.join(P, f"{TAG}_{stem}_pmc_traffic.txt")
    if not os.path.exists(f):
        continue
    c, kern = read(f)
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        print("incomplete:", f)
        continue
    e = traffic.setdefault(model, {}).setdefault(str(n), {}).setdefault(dt, {})
    if slot == "single":
        e.update({"fetch_kib": c["FETCH_SIZE"][0], "write_kib": c["WRITE_SIZE"][0], "source": rel(f), "kernel": kern})
    else:
        r = e.setdefault("rings", {})
        src = set(filter(None, [s.strip() for s in r.get("source", "").split(",") if TAG in s]))
        src.add(rel(f) + f" ({K}-step launch)")
        r["source"] = ", ".join(sorted(src))
        r["kernel"] = kern
        if slot in ("rings_short", "rings_both"):
            r["fetch_kib_per_launch"] = c["FETCH_SIZE"][0] if slot == "rings_short" else c["FETCH_SIZE"][0] / K * 20
            r["write_kib_per_step_short"] = c["WRITE_SIZE"][2]
        if slot in ("rings_long", "rings_both"):
            r["fetch_kib_per_step_long"] = c["FETCH_SIZE"][2]
            r["write_kib_per_step_long"] = c["WRITE_SIZE"][2]
    print(f"traffic  {model} x {n} [{dt}] {slot}: fetch {c['FETCH_SIZE'][0]:.1f} KiB write {c['WRITE_SIZE'][0]:.1f} KiB per launch ({kern})")

SQ_CASES = [("ant4096_f64_sq_counters_loop", "ant", 4096, "f64", 1000), ("ant8192_f64_sq_counters", "ant", 8192, "f64", 500),
            ("laikago_soft8192_f64_sq_counters", "laikago_soft", 8192, "f64", 0),
            ("laikago_soft8192_f64_loop_sq_counters", "laikago_soft", 8192, "f64", 500),
            ("laikago_soft4096_f64_loop_sq_counters", "laikago_soft", 4096, "f64", 500),
            ("pendulum5_4096_f32rec_sq_counters", "pendulum5", 4096, "f32", 500)]
for stem, model, n, dt, K in SQ_CASES:
    f = os.path.join(P, f"{TAG}_{stem}.txt")
    if not os.path.exists(f):
        continue
    c, kern = read(f)
    if "SQ_INSTS_VALU" not in c:
        print("incomplete:", f)
        continue
    valu, us, _ = c["SQ_INSTS_VALU"]
    steps = max(K, 1)
    e = {"kernel": kern, "source": rel(f), "steps_per_launch": steps, "launch_us": us,
         "valu_insts_per_env_step": valu / (n * steps),
         "valu_issue_frac": valu * 4.0 / (SIMDS * us * PEAK_MHZ)}
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        e["wait_any_frac"] = c["SQ_WAIT_ANY"][0] / c["SQ_WAVE_CYCLES"][0]
    if "SQ_INSTS_LDS" in c:
        e["lds_insts_per_env_step"] = c["SQ_INSTS_LDS"][0] / (n * steps)
    if "SQ_INSTS_SALU" in c:
        e["salu_insts_per_env_step"] = c["SQ_INSTS_SALU"][0] / (n * steps)
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_ACTIVE_INST_LDS" in c and c["SQ_ACTIVE_INST_LDS"][0] > 0:
        e["lds_bank_conflict_over_active_lds"] = c["SQ_LDS_BANK_CONFLICT"][0] / c["SQ_ACTIVE_INST_LDS"][0]
    sq.setdefault(model, {}).setdefault(str(n), {})[dt] = e
    print(f"sq       {model} x {n} [{dt}]: {e['valu_insts_per_env_step']:.0f} VALU / env-step, issue {e['valu_issue_frac']:.3f} ({kern})")

json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
json.dump(sq, open(sq_path, "w"), indent=1)
