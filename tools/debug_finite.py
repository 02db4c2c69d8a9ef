"""Debug aid: replay bench.py's closed loop and report the first step / environment whose y turns
non-finite; dumps that environment's input record of the offending step to gpurun_out/nonfinite.npz
(then:  oraclelib.step(model, x)  on the host tells whether the reference diverges there too)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tds_amd  # noqa: E402
from tds_amd import hip_backend  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="laikago")
ap.add_argument("--envs", type=int, default=8192)
ap.add_argument("--steps", type=int, default=1100)
a = ap.parse_args()
m = tds_amd.load_model(a.model)
n = a.envs
sim = hip_backend.HipSim(m, n, dtype="f64")
nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
rng = np.random.default_rng(3)
x0 = np.zeros((n, m.input_dim))
ip = np.array([m.initial_poses[i] for i in range(adim)])
x0[:, 2] = 0.48
x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
x0[:, -3:] = [15, 0.3, 3] if a.model.startswith("ant") else [100, 2, 50]
sim.x.copy_(torch.from_numpy(x0).cuda())
for _ in range(10):
    sim.step(None)
pool = 16
actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (pool, n, adim))).cuda().contiguous()
obs = torch.zeros((n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
big = []
hist = []  # inputs of the last 60 steps (all environments)
for i in range(a.steps):
    xprev = sim.x.clone()
    xprev[:, nq + nd:nq + nd + adim] = actions[i % pool]
    hist.append(xprev)
    if len(hist) > 60:
        hist.pop(0)
    sim.step(actions[i % pool], 1, obs)
    bad = ~torch.isfinite(sim.y).all(dim=1)
    mx = sim.y[:, :nq + nd].abs().max().item()
    big.append(mx)
    if bad.any():
        e = int(torch.nonzero(bad)[0].item())
        xin = xprev[e].cpu().numpy()
        xin[nq + nd:nq + nd + adim] = actions[i % pool][e].cpu().numpy()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "nonfinite.npz"), x=xin, y=sim.y[e].cpu().numpy(), step=i, env=e,
                 hist=torch.stack([h[e] for h in hist]).cpu().numpy())
        print(f"first non-finite: step {i} env {e} ({int(bad.sum())} envs); |state| max before: {big[-5:]}")
        print("x =", np.array2string(xin, precision=4, max_line_width=200))
        break
else:
    print("all finite; max |q,qd| over the run:", max(big))
