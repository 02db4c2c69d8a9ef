#!/usr/bin/env python3
"""How long do Laikago environments stay up (and finite) under uniform random actions of a given amplitude, no resets?
(The reference has no joint limits: a fallen robot driven by random actions can blow up numerically.)
usage: python tools/laikago_stability.py [model=laikago_soft] [n=8192] [steps=1200]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend

name = sys.argv[1] if len(sys.argv) > 1 else "laikago_soft"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
m = tds_amd.load_model(name)
nq, adim = m.dof_q, m.action_dim
for amp in (0.4, 0.2, 0.1, 0.05):
    rng = np.random.default_rng(3)
    x0 = np.zeros((n, m.input_dim))
    ip = np.array([m.initial_poses[i] for i in range(adim)])
    x0[:, 2] = 0.48
    x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
    x0[:, -3:] = [15, 0.3, 3] if name.startswith("ant") else [100, 2, 50]
    sim = hip_backend.HipSim(m, n)
    sim.x.copy_(torch.from_numpy(x0).cuda())
    for _ in range(10):
        sim.step(None)
    acts = torch.from_numpy(rng.uniform(-amp, amp, (16, n, adim))).cuda().contiguous()
    obs = torch.zeros((n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
    out = []
    for k in range(0, steps, 200):
        sim.step_many(acts, 200, obs, first_block=k % 16)
        torch.cuda.synchronize()
        bad = int((~torch.isfinite(sim.y).all(dim=1)).sum())
        done = int((obs[:, -1] != 0).sum())
        out.append(f"{k + 200}: nonfinite {bad} done {done}")
    print(f"{name} x{n} amp {amp}: " + " | ".join(out))
