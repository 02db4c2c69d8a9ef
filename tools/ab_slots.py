#!/usr/bin/env python3
"""Same-process, same-box A/B of the experiment slots of the library (csrc/tds_kernels.h: EXPERIMENT SLOTS; built by
tools/build_alt.sh): one handle per slot (option alt_build = k; 0 = the library's own kernels), the timed launches of
bench.py (tds_hip_step_many_rings: per-step y and obs records into 64-slot rings) interleaved round-robin so that clock
and thermal drift hit every slot alike; every slot's records are checked against slot 0's (max relative difference).
usage: python tools/ab_slots.py [--model ant] [--envs 4096] [--slots 0,1,2,...] [--steps 1000] [--short 20] [--reps 5]
       [--opt key=value ...]   (library options of every handle, e.g. loop_w2=0 for the one-wave loop build)"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="ant")
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--slots", default="0,1,2,3,4,5,6")
ap.add_argument("--steps", type=int, default=1000)
ap.add_argument("--short", type=int, default=20)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--opt", action="append", default=[])
args = ap.parse_args()

m = tds_amd.load_model(args.model)
n = args.envs
rng = np.random.default_rng(0)
x0 = np.zeros((n, m.input_dim))
nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION:  # (bench.py's start state)
    ip = np.array([m.initial_poses[i] for i in range(adim)])
    x0[:, 2] = 0.48
    x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
    x0[:, -3:] = [15, 0.3, 3] if args.model.startswith("ant") else [100, 2, 50]
else:
    x0[:, :nq] = rng.uniform(-1, 1, (n, nq))
amp = 0.4 if args.model.startswith("ant") else 0.1
acts = torch.from_numpy(rng.uniform(-amp, amp, (16, n, adim))).cuda().contiguous()
base_opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.opt}
RS = 64
per_line = 16
ystr = -(-m.output_dim // per_line) * per_line
slots = [int(s) for s in args.slots.split(",")]
sims = {}
for k in slots:
    o = dict(base_opts)
    if k:
        o["alt_build"] = k
    try:
        s = hip_backend.HipSim(m, n, dtype="f64", options=o)
        s.x.copy_(torch.from_numpy(x0).cuda())
        obs = torch.zeros((RS, n, s.obs_dim + 2), dtype=torch.float64, device="cuda")
        yr = torch.zeros((RS, n, ystr), dtype=torch.float64, device="cuda")
        for _ in range(10):
            s.step(None)
        s.step_many_rings(acts, 64, obs, yr)  # (also the first use of the slot: refused here if it does not exist)
        torch.cuda.synchronize()
        sims[k] = (s, obs, yr)
    except Exception as e:  # noqa: BLE001
        print(f"slot {k}: not usable ({e})")
# parity of the slots among themselves: 40 steps from the same SETTLED state (the first slot's, after its 10 settle + 64
# steps: robots on the ground, contacts in play)
ref = None
xs_common = sims[slots[0]][0].x.clone()
for k, (s, obs, yr) in sims.items():
    s.x.copy_(xs_common)
    s.step_many_rings(acts, 40, obs, yr)
    torch.cuda.synchronize()
    y = yr[39][:, :m.output_dim].cpu().numpy()
    if ref is None:
        ref = y
    d = float(np.max(np.abs(y - ref) / np.maximum(np.abs(ref), 1e-3)))
    print(f"slot {k}: 40 steps, max rel difference to slot {slots[0]}: {d:.3e}, finite {bool(np.isfinite(y).all())}")
# spin-up (clocks)
for k, (s, obs, yr) in sims.items():
    s.step_many_rings(acts, args.steps, obs, yr)
torch.cuda.synchronize()
# (every timed launch runs right behind the SAME conditioning launch — 256 steps of the first slot's handle: a launch that
#  follows a slow one inherits its clocks, up to 10 % at these kernel sizes; the order of the slots rotates from repetition
#  to repetition)
# (a handle of its own: a graph-form model keeps ONE graph per handle — alternating 256- and K-step calls on a timed handle
#  would re-instantiate its graph inside the timed region)
cs = hip_backend.HipSim(m, n, dtype="f64", options=dict(base_opts))
cs.x.copy_(sims[slots[0]][0].x)
cobs, cyr = torch.zeros_like(sims[slots[0]][1]), torch.zeros_like(sims[slots[0]][2])
cond = cs.prepared_step_many_rings(acts, 256, cobs, cyr)
for K in (args.steps, args.short):
    calls = {k: s.prepared_step_many_rings(acts, K, obs, yr) for k, (s, obs, yr) in sims.items()}
    t = {k: [] for k in sims}
    order = list(sims)
    for rep in range(args.reps + 1):
        order = order[1:] + order[:1]
        for k in order:
            cond()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            calls[k]()
            torch.cuda.synchronize()
            if rep:
                t[k].append((time.perf_counter() - t0) / K * 1e6)
    print(f"--- {args.model} x {n}, {K}-step launches, us per step (min / median of {args.reps}), {base_opts or ''}")
    b = min(t[slots[0]]) if slots[0] in t else None
    for k in sims:
        print(f"slot {k}: {min(t[k]):7.3f} / {sorted(t[k])[len(t[k]) // 2]:7.3f}" + (f"   {min(t[k]) / b:6.3f} x slot {slots[0]}" if b else ""))
