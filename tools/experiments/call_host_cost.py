#!/usr/bin/env python3
"""Host-side cost of ONE tds_hip_step_many_rings call (what sits between the driver's opening synchronisation and the kernel's
first instruction): wall time of the prepared call returning, of event records, and of the whole synchronised region.
usage: python tools/experiments/call_host_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tds_amd
from tds_amd import hip_backend

m = tds_amd.load_model("ant"); n = 4096
sim = hip_backend.HipSim(m, n)
rng = np.random.default_rng(1)
x0 = np.zeros((n, m.input_dim)); ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
x0[:, 2] = 0.48; x0[:, 6:14] = ip; x0[:, -3:] = [15, 0.3, 3]
sim.x.copy_(torch.from_numpy(x0).cuda())
actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, 8))).cuda().contiguous()
obs_ring = torch.zeros((64, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
y_ring = torch.zeros((64, n, 160), dtype=torch.float64, device="cuda")
K = 20
call = sim.prepared_step_many_rings(actions, K, obs_ring, y_ring, first_block=0, obs_first=0, y_first=0)
for _ in range(20):
    call()
torch.cuda.synchronize()
enq, reg, regev = [], [], []
for rep in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); call(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    enq.append(t1 - t0); reg.append(t2 - t0)
for rep in range(200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record(); call(); e1.record()
    while not e1.query():
        pass
    torch.cuda.synchronize(); t2 = time.perf_counter()
    regev.append((t2 - t0, e0.elapsed_time(e1) * 1e-3))
med = lambda a: 1e6 * float(np.median(a))
print(f"ant x {n}, {K}-step ring call (median of 200): the call returns after {med(enq):.1f} us; call + synchronize {med(reg):.1f} us; "
      f"with an event pair around it (bench.py's timed region) {med([a for a, b in regev]):.1f} us, of which between the events {med([b for a, b in regev]):.1f} us")
