"""Runs ON THE GPU BOX: a K-step tds_hip_step_many call (K = 20, 100, 1000; Ant x 4096, one step-loop launch) repeated back
to back, wall time and HIP-event time, by the way the host waits for it: torch.cuda.synchronize alone, after
hipStreamSynchronize, after an event wait, after polling the event."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import tds_amd
from tds_amd import hip_backend
m = tds_amd.load_model("ant"); n = 4096
rng = np.random.default_rng(3)
nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
x0 = np.zeros((n, m.input_dim)); x0[:, 2] = 0.48
x0[:, 6:nq] = np.array([m.initial_poses[i] for i in range(adim)]) + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
x0[:, -3:] = [15, 0.3, 3]
a = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, adim))).cuda().contiguous()
sim = hip_backend.HipSim(m, n)
sim.x.copy_(torch.from_numpy(x0).cuda())
obs = torch.zeros((n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
for _ in range(10):
    sim.step(None)
def wait(form, ev):
    if form == "stream":
        sim.sync()
    elif form == "event":
        ev.synchronize()
    elif form == "poll":
        while not ev.query():
            pass
    torch.cuda.synchronize()
for K in (20, 100, 1000):
    sim.step_many(a, K, obs); torch.cuda.synchronize()
    for form in ("torch", "stream", "event", "poll"):
        ts = []
        for rep in range(15):
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record()
            sim.step_many(a, K, obs)
            ev1.record()
            wait(form, ev1)
            ts.append(time.perf_counter() - t0)
        gpu = ev0.elapsed_time(ev1) * 1e3
        print(f"K={K} wait={form}: wall median {np.median(ts)*1e6:.0f} us = {np.median(ts)*1e6/K:.2f} us/step (min {np.min(ts)*1e6:.0f}); events {gpu:.0f} us")
