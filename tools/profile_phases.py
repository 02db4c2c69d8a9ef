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
