#!/usr/bin/env python3
"""What runs right in front of bench.py's 20-step timed region, and what the region then costs (median of 60):
  A  a 256-step launch of a scratch handle WITHOUT record rings (bench.py's spin-up through r06d)
  B  the same launch WITH rings: the scratch handle's records go into the slots the timed steps are about to overwrite
  C  nothing but the synchronisation
  D  the timed handle's previous 20-step call (a tight loop of the timed call)
usage: python tools/experiments/region_front.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tds_amd
from tds_amd import hip_backend

m = tds_amd.load_model("ant"); n = 4096; K = 20; RS = 64
sim, scratch = hip_backend.HipSim(m, n), hip_backend.HipSim(m, n)
rng = np.random.default_rng(1)
x0 = np.zeros((n, m.input_dim)); ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
x0[:, 2] = 0.48; x0[:, 6:14] = ip + 0.05 * rng.uniform(-1, 1, (n, 8)); x0[:, -3:] = [15, 0.3, 3]
for s in (sim, scratch):
    s.x.copy_(torch.from_numpy(x0).cuda())
actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, 8))).cuda().contiguous()
obs_ring = torch.zeros((RS, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
y_ring = torch.zeros((RS, n, 160), dtype=torch.float64, device="cuda")
for _ in range(12):
    scratch.step_many(actions, 500)
torch.cuda.synchronize()
res = {}
i = 0
for var in "ABCDABCD":
    out = res.setdefault(var, [])
    for rep in range(30):
        call = sim.prepared_step_many_rings(actions, K, obs_ring, y_ring, first_block=i % 16, obs_first=i % RS, y_first=i % RS)
        if var == "A":
            scratch.step_many(actions, 256)
        elif var == "B":
            scratch.step_many_rings(actions, 256, obs_ring, y_ring, first_block=i % 16, obs_first=i % RS, y_first=i % RS)
        if var in "AB":
            ev = torch.cuda.Event(); ev.record()
            while not ev.query():
                pass
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); call(); e1.record()
        while not e1.query():
            pass
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out.append((t1 - t0, e0.elapsed_time(e1) * 1e-3))
        i += K
        if var != "D":
            time.sleep(0.002)  # (bench.py marshals the call on the host before the region: the GPU idles)
med = lambda a: 1e6 * float(np.median(a))
for var in "ABCD":
    print(f"{var}: region {med([a for a, b in res[var]]):.1f} us, between the events {med([b for a, b in res[var]]):.1f} us")
