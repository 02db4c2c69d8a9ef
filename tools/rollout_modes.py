#!/usr/bin/env python3
"""Runs ON THE GPU BOX: throughput of tds_hip_rollout (Ant, 100 policy steps) in its two forms — one launch of the
step-loop build / one straight-line step launch per step + policy-and-bookkeeping kernel — per batch size."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tds_amd  # noqa: E402
from tds_amd import hip_backend  # noqa: E402

m = tds_amd.load_model(sys.argv[1] if len(sys.argv) > 1 else "ant")
nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
od = nq + nd
R = 100
for n in (4096, 8192, 16384, 32768):
    rng = np.random.default_rng(3)
    x0 = np.zeros((n, m.input_dim))
    ip = np.array([m.initial_poses[i] for i in range(adim)])
    x0[:, 2] = 0.48
    x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
    x0[:, -3:] = [15, 0.3, 3]
    pol = torch.from_numpy(rng.normal(0.0, 0.05, (n, adim * od + adim))).cuda().contiguous()
    out = {}
    for mode in ("single", "per_step", None):
        sim = hip_backend.HipSim(m, n)
        sim.x.copy_(torch.from_numpy(x0).cuda())
        for _ in range(10):
            sim.step(None)
        xs = sim.x.clone()
        ret, cnt = sim.rollout(pol, R, 0.0, mode=mode)
        torch.cuda.synchronize()
        out[mode] = (ret.cpu().numpy(), cnt.cpu().numpy())
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            sim.rollout(pol, R, 0.0, mode=mode)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{m.name.decode()} x{n} mode={mode}: {n * R * reps / dt:.4g} env-steps/s", flush=True)
        sim.close()
    # agreement of the two forms on a SHORT rollout (over 100 steps of contact dynamics under random policies the
    # 1e-16 differences of the policy's summation order grow chaotically, incl. different done times)
    short = {}
    for mode in ("single", "per_step"):
        sim = hip_backend.HipSim(m, n)
        sim.x.copy_(xs)
        r8, c8 = sim.rollout(pol, 8, 0.0, mode=mode)
        short[mode] = (r8.cpu().numpy(), c8.cpu().numpy())
        sim.close()
    a, b = short["single"], short["per_step"]
    same = np.array_equal(a[1], b[1])
    live = np.isfinite(a[0]) & np.isfinite(b[0])
    err = np.max(np.abs(a[0][live] - b[0][live]) / np.maximum(np.abs(a[0][live]), 1e-3))
    print(f"  single vs per_step over 8 steps: step counts equal {same}, returns max rel diff {err:.2e}")
