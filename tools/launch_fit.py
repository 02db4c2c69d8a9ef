"""Kernel time of ONE ring launch (two-wavefront step-loop build, records of every step) against its step count K, from
HIP events around the launch: T(K) = a + b K.  a = what a launch costs before / after its steps (dispatch, prologue, first
pass through the code, tail skew); b = the steady per-step time.  Back-to-back launches and launches behind an idle gap."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tds_amd
from tds_amd import hip_backend

model, n = (sys.argv[1] if len(sys.argv) > 1 else "ant"), int(sys.argv[2]) if len(sys.argv) > 2 else 4096
m = tds_amd.load_model(model)
rng = np.random.default_rng(3)
nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
x0 = np.zeros((n, m.input_dim))
loco = m.step_mode == tds_amd.TDS_STEP_LOCOMOTION
if loco:
    x0[:, 2] = 0.48
    x0[:, 6:nq] = np.array([m.initial_poses[i] for i in range(adim)]) + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
    x0[:, -3:] = [15, 0.3, 3] if model == "ant" else [100, 2, 50]
else:
    x0[:, :nq] = rng.uniform(-1, 1, (n, nq))
a = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, adim)) * (1.0 if loco else 0.0)).cuda().contiguous()
sim = hip_backend.HipSim(m, n)
sim.x.copy_(torch.from_numpy(x0).cuda())
S = 64
obs_ring = torch.zeros((S, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
y_ring = torch.zeros((S, n, sim.output_dim), dtype=torch.float64, device="cuda")
for _ in range(20):
    sim.step(None)
sim.step_many_rings(a, 200, obs_ring, y_ring); torch.cuda.synchronize()
Ks = [1, 2, 3, 5, 10, 20, 40, 64]
for gap in ("back to back", "after 2 ms of idle"):
    res = {}
    for K in Ks:
        sim.step_many_rings(a, K, obs_ring, y_ring, prepare_only=True)
        ts = []
        for rep in range(9):
            if gap != "back to back":
                torch.cuda.synchronize(); time.sleep(0.002)
            else:
                sim.step_many_rings(a, K, obs_ring, y_ring)   # the launch in front of the timed one
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sim.step_many_rings(a, K, obs_ring, y_ring)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[K] = float(np.median(ts))
    A = np.array([[1.0, k] for k in Ks]); bvec = np.array([res[k] for k in Ks])
    (a0, b0), *_ = np.linalg.lstsq(A, bvec, rcond=None)
    print(f"{model} x{n}, {gap}: " + ", ".join(f"K={k}: {res[k]:.1f}" for k in Ks) + f" us  ->  T(K) = {a0:.1f} + {b0:.2f} K us")
