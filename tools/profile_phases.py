#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of one step (workgroup 0), for DESIGN.md / tuning.
The environments are brought to the same kind of state bench.py measures on: reset-like state,
10 settle steps, then `--steps` closed-loop steps with fresh random actions.
usage: python tools/profile_phases.py [model] [n_envs] [lanes] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend

name = sys.argv[1] if len(sys.argv) > 1 else "ant"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lanes = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
m = tds_amd.load_model(name)
nq, adim = m.dof_q, m.action_dim
rng = np.random.default_rng(3)
x0 = np.zeros((n, m.input_dim))
loco = m.step_mode == tds_amd.TDS_STEP_LOCOMOTION
if loco:
    ip = np.array([m.initial_poses[i] for i in range(adim)])
    x0[:, 2] = 0.48
    x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
    x0[:, -3:] = [15, 0.3, 3] if name.startswith("ant") else [100, 2, 50]
else:
    x0[:, :nq] = rng.uniform(-1, 1, (n, nq))
sim = hip_backend.HipSim(m, n, lanes_per_env=lanes)
sim.x.copy_(torch.from_numpy(x0).cuda())
for _ in range(10):
    sim.step(None)
amp = 0.4 if loco else 0.0
for _ in range(steps):
    sim.step(torch.from_numpy(rng.uniform(-amp, amp, (n, adim))).cuda().contiguous())
torch.cuda.synchronize()
x_prof = sim.x.clone()
for _ in range(3):
    sim.x.copy_(x_prof)
    ph = sim.profile_phases()
tot = sum(ph.values())
info = sim.kernel_info()
print(f"{name} n={n} after {steps} closed-loop steps {info} total cycles (wg 0): {tot}")
for k, v in ph.items():
    print(f"  {k:22s} {v:8d}  {100.0 * v / tot:5.1f}%")
if m.has_plane:
    # penetrating contact points per environment / per wavefront in the profiled state (from the oracle's
    # narrowphase on a sample), to relate K / L cycles to the constraint-row count
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import oraclelib
        xs = x_prof[:256].cpu().numpy()
        na = np.array([int((oraclelib.step_debug(m, xs[i])["contacts"][:, 9] < 0).sum()) for i in range(xs.shape[0])])
        epb = info["envs_per_block"]
        wmax = na.reshape(-1, epb).max(axis=1)
        print(f"  penetrating contacts/env: mean {na.mean():.2f} max {na.max()}  per-wave max: mean {wmax.mean():.2f}"
              f"  wave 0: {wmax[0]}")
    except Exception as e:  # diagnostic only
        print("  (contact statistics unavailable:", e, ")")
tw = sim.profile_phases_two_waves()
if tw is not None:
    MAIN = ["start", "A constants, load, PD", "B jcalc", "C kinematics  (-> barrier 1)", "D inertias  (after barrier 1)",
            "E composite sweep", "G mass matrix", "H LDLt  (-> barrier 2)", "F solve  (after barrier 2; -> barrier 3)",
            "after barrier 3", "(not stamped in this form)", "right-hand sides", "L contact solve", "M/N integrate + pack"]
    HELP = ["start", "constants loaded  (-> barrier 1)", "after barrier 1", "I narrowphase", "M1 visual poses + y tail",
            "J rows  (-> barrier 2 / counts published)", "after barrier 2 / publishing", "K row solves  (-> barrier 3)", "after barrier 3"]
    print("two-wavefront form, workgroup 0: cycle at which each phase ENDS (0 = the main wavefront's first stamp)")
    ex = tw[2]
    print(f"  workgroup 0 first -> last stamp: {tw[0][13]} shader cycles in {ex['wg0_wall_us']:.2f} us of the 100 MHz wall clock"
          f" = {tw[0][13] / max(ex['wg0_wall_us'], 1e-9):.0f} MHz; last workgroup: starts at cycle {ex['last_wg_start']}, "
          f"ends at {ex['last_wg_end']} ({ex['last_wg_end_wall_us']:.2f} us after workgroup 0's start)")
    ws, we = np.array(ex["wg_start_us"]), np.array(ex["wg_end_us"])
    dur = we - ws
    print(f"  all {len(ws)} workgroups (100 MHz wall clock, us relative to workgroup 0's first stamp): first start {ws.min():.2f}, "
          f"last start {ws.max():.2f}, first end {we.min():.2f}, last end {we.max():.2f}; duration min {dur.min():.2f} "
          f"mean {dur.mean():.2f} max {dur.max():.2f}; span first start -> last end {we.max() - ws.min():.2f}")
    print("  duration percentiles 5/25/50/75/95/99:", np.round(np.percentile(dur, [5, 25, 50, 75, 95, 99]), 2),
          " start percentiles:", np.round(np.percentile(ws, [5, 25, 50, 75, 95, 99]), 2))
    for title, names, st in (("  main wavefront", MAIN, tw[0]), ("  helper wavefront", HELP, tw[1])):
        print(title)
        prev = st[0]
        for k, v in zip(names, st):
            if abs(v) > 10**9:  # (a stamp this form does not take)
                continue
            print(f"    {k:42s} {v:8d}  (+{v - prev})")
            prev = v
