"""Kernel duration of ONE step-loop launch with per-step records (tds_hip_step_many_rings, Ant x 4096) against its number of
steps K: duration(K) = a + b K — what of a short launch is fixed cost (prologue, epilogue, the slowest workgroup's tail)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tds_amd
from tds_amd import hip_backend

m = tds_amd.load_model("ant")
n = 4096
rng = np.random.default_rng(3)
nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
x0 = np.zeros((n, m.input_dim)); x0[:, 2] = 0.48
x0[:, 6:nq] = np.array([m.initial_poses[i] for i in range(adim)]) + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
x0[:, -3:] = [15, 0.3, 3]
a = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, adim))).cuda().contiguous()
sim = hip_backend.HipSim(m, n)
sim.x.copy_(torch.from_numpy(x0).cuda())
sim.set_timing(True)
RS = 64
obs_ring = torch.zeros((RS, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
y_ring = torch.zeros((RS, n, 160), dtype=torch.float64, device="cuda")
sim.step_many_rings(a, 500, obs_ring, y_ring)
torch.cuda.synchronize()
Ks, Ds = [], []
for K in (1, 2, 3, 5, 10, 20, 40, 80, 160, 320, 1000):
    d = []
    for _ in range(7):
        sim.step_many_rings(a, 256, obs_ring, y_ring)  # (clocks, caches: as behind bench.py's scratch steps)
        sim.step_many_rings(a, K, obs_ring, y_ring)
        torch.cuda.synchronize()
        d.append(sim.last_kernel_ms() * 1e3)
    Ks.append(K); Ds.append(float(np.median(d)))
    print(f"K {K:5d}: {Ds[-1]:9.1f} us = {Ds[-1] / K:7.2f} us/step")
b, a0 = np.polyfit(Ks[-4:], Ds[-4:], 1)
print(f"fit over K >= 160: {a0:.1f} us + {b:.3f} us/step;  excess of the short launches over b K: " +
      ", ".join(f"K={k}: {d - b * k:.1f}" for k, d in zip(Ks[:7], Ds[:7])))
