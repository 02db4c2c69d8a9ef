"""Which workgroups of the two-wavefront step kernel are the slow ones?  Per-workgroup wall-clock durations
(PROF build) against the block index, XCD (block % 8), start time and the wavefront's contact rows."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import tds_amd
from tds_amd import hip_backend
import oraclelib

name, n = "ant", 4096
m = tds_amd.load_model(name)
sim = hip_backend.HipSim(m, n)
rng = np.random.default_rng(3)
nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
x0 = np.zeros((n, m.input_dim))
x0[:, 2] = 0.48
x0[:, 6:nq] = np.array([m.initial_poses[i] for i in range(adim)]) + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
x0[:, -3:] = [15, 0.3, 3]
sim.x.copy_(torch.from_numpy(x0).cuda())
for _ in range(10):
    sim.step(None)
for t in range(100):
    sim.step(torch.from_numpy(rng.uniform(-0.4, 0.4, (n, adim))).cuda())
xs = sim.x.clone()
na = np.array([int((oraclelib.step_debug(m, xs[i].cpu().numpy())["contacts"][:, 9] < 0).sum()) for i in range(n)])
NA = na.reshape(-1, 4).max(axis=1)
durs, starts = [], []
for rep in range(6):
    sim.x.copy_(xs)
    tw = sim.profile_phases_two_waves()
    ex = tw[2]
    ws, we = np.array(ex["wg_start_us"]), np.array(ex["wg_end_us"])
    durs.append(we - ws); starts.append(ws)
durs, starts = np.array(durs), np.array(starts)
d = durs[1:].mean(axis=0)
print("mean duration by wavefront contact slots NA:", {int(k): round(float(d[NA == k].mean()), 2) for k in np.unique(NA)},
      "counts", {int(k): int((NA == k).sum()) for k in np.unique(NA)})
print("mean duration by XCD (block % 8):", np.round([d[i::8].mean() for i in range(8)], 2))
print("mean start by XCD:", np.round([starts[1:].mean(axis=0)[i::8].mean() for i in range(8)], 2))
for rep in range(1, 6):
    dd = durs[rep]
    slow = np.argsort(dd)[-12:][::-1]
    print(f"rep {rep}: slowest:", [(int(b), int(b % 8), int(NA[b]), round(float(starts[rep][b]), 2), round(float(dd[b]), 2)) for b in slow])
print("correlation of a block's duration between repetitions:", np.round(np.corrcoef(durs[1:])[0], 2))
# residual after removing the NA effect: same blocks slow every time?
res = durs[1:] - np.array([[durs[1:][r][NA == NA[b]].mean() for b in range(len(NA))] for r in range(5)])
print("blocks slow (> +1.5 us over their NA class) in >= 4 of 5 reps:", np.nonzero((res > 1.5).sum(axis=0) >= 4)[0][:40])
print("slow-count histogram over blocks:", np.bincount((res > 1.5).sum(axis=0)))
