"""Fixed cost of one tds_hip_step_many call (graph launches + fork / join + final synchronisation) against its K steps:
wall time of a K-step call bracketed by synchronisations, for K = 20 and 1000, chains 1 and 2."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import tds_amd
from tds_amd import hip_backend
from test_multi_gpu import _start

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
    for _ in range(10):
        sim.step(None)
    res = {}
    for K in (1000, 20):
        sim.step_many_prepare(a, K, obs)
        sim.step_many(a, K, obs); torch.cuda.synchronize()
        ts = []
        for rep in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sim.step_many(a, K, obs)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[K] = np.median(ts) * 1e6
    per = res[1000] / 1000
    print(f"chains {chains}: 1000 steps {res[1000]:.0f} us ({per:.2f} us/step); 20 steps {res[20]:.0f} us "
          f"({res[20] / 20:.2f} us/step) -> fixed cost of a call {res[20] - 20 * per:.0f} us")
