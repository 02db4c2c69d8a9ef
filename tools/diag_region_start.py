#!/usr/bin/env python3
"""What does the kernel of a 20-step timed region pay for what ran in front of the region?  One 20-step ring launch of the
timed handle (HIP events around it AND the host clock around launch + polled wait, as bench.py times it), behind:
  a  the same launch, no synchronisation in between (back to back)
  b  256 ring steps of the SAME handle, polled wait, torch.cuda.synchronize()
  c  256 plain steps of a scratch handle (bench.py's keep-warm launch), polled wait, synchronize
  d  256 ring steps of a scratch handle with rings of its own, polled wait, synchronize
  e  blocking synchronize + 2 ms of sleep
usage: python tools/diag_region_start.py [model] [envs] [K]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tds_amd
from tds_amd import hip_backend

model = sys.argv[1] if len(sys.argv) > 1 else "ant"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
m = tds_amd.load_model(model)
rng = np.random.default_rng(3)
nq, adim = m.dof_q, m.action_dim
x0 = np.zeros((n, m.input_dim))
x0[:, 2] = 0.48
x0[:, 6:nq] = np.array([m.initial_poses[i] for i in range(adim)]) + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
x0[:, -3:] = [15, 0.3, 3] if model.startswith("ant") else [100, 2, 50]
a = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, adim))).cuda().contiguous()
S = 64
ystr = -(-m.output_dim // 16) * 16


def handle():
    s = hip_backend.HipSim(m, n)
    s.x.copy_(torch.from_numpy(x0).cuda())
    o = torch.zeros((S, n, s.obs_dim + 2), dtype=torch.float64, device="cuda")
    y = torch.zeros((S, n, ystr), dtype=torch.float64, device="cuda")
    for _ in range(20):
        s.step(None)
    s.step_many_rings(a, 300, o, y)
    return s, o, y


sim, o, y = handle()
scr, so, sy = handle()
torch.cuda.synchronize()
timed = sim.prepared_step_many_rings(a, K, o, y)
same256 = sim.prepared_step_many_rings(a, 256, o, y)
scr_rings256 = scr.prepared_step_many_rings(a, 256, so, sy)
scr.step_many(a, 256)
torch.cuda.synchronize()


def polled_sync():
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():
        pass
    torch.cuda.synchronize()


def front(kind):
    if kind == "a":
        timed()
    elif kind == "b":
        same256(); polled_sync()
    elif kind == "c":
        scr.step_many(a, 256); polled_sync()
    elif kind == "d":
        scr_rings256(); polled_sync()
    else:
        torch.cuda.synchronize(); time.sleep(0.002)


names = {"a": "back to back", "b": "same handle, 256 ring steps, polled sync", "c": "scratch handle, 256 plain steps, polled sync (bench.py)",
         "d": "scratch handle, 256 ring steps, polled sync", "e": "blocking sync + 2 ms idle"}
res = {k: ([], []) for k in names}
for rep in range(8):
    for kind in names:
        front(kind)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        timed()
        e1.record()
        while not e1.query():
            pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if rep:
            res[kind][0].append(e0.elapsed_time(e1) * 1e3)
            res[kind][1].append((t1 - t0) * 1e6)
print(f"{model} x {n}, one {K}-step ring launch: kernel (events) / region (host clock), us, median of 7 [min]")
for kind, nm in names.items():
    ke, ho = res[kind]
    print(f"  {kind}  {nm:58s} {np.median(ke):7.1f} [{min(ke):6.1f}] / {np.median(ho):7.1f} [{min(ho):6.1f}]   -> {n * K / np.median(ho) * 1e6:.3e} env-steps/s")
