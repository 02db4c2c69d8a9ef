#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of one step (workgroup 0), for DESIGN.md / tuning.
usage: python tools/profile_phases.py [model] [n_envs] [lanes]"""
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
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else None
m = tds_amd.load_model(name)
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
rng = np.random.default_rng(0)
x = g["x"][rng.integers(0, g["x"].shape[0], n)]
sim = hip_backend.HipSim(m, n, lanes_per_env=lanes)
sim.x.copy_(torch.from_numpy(x).cuda())
for _ in range(3):
    ph = sim.profile_phases()
tot = sum(ph.values())
print(f"{name} n={n} {sim.kernel_info()} total cycles (wg 0): {tot}")
for k, v in ph.items():
    print(f"  {k:22s} {v:8d}  {100.0 * v / tot:5.1f}%")
