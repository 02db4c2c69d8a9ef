"""First launch of a freshly prepared step_many graph against later launches (the driver's 20-step run times a first launch)."""
import os, sys, time
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
for chains in (1, 2):
    sim = hip_backend.HipSim(m, n)
    sim.x.copy_(torch.from_numpy(x0).cuda())
    sim.set_graph_chains(chains)
    obs = torch.zeros((n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
    sim.step_many(a, 5, obs); torch.cuda.synchronize()   # warm-up with another graph, as bench.py does
    out = []
    for first in (3, 7, 11):  # three fresh graphs (the cache is keyed by first_block)
        sim.step_many_prepare(a, 20, obs, first_block=first)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            sim.step_many(a, 20, obs, first_block=first)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
        out.append(ts)
    print(f"chains {chains}: 20-step call, us (1st, 2nd, 3rd launch of a fresh graph):", [[round(t) for t in ts] for ts in out])
