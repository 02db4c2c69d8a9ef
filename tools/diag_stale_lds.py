#!/usr/bin/env python3
"""Diagnostic for tests/test_hip_parity.py::test_no_step_reads_stale_lds[loop-*]: which entries of the y record differ
between a clean 3-substep launch and one after the LDS has been poisoned — or between two clean ones.
usage: python tools/diag_stale_lds.py [model ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend

names = sys.argv[1:] or ["sphere_spherical", "humanoid_spherical", "sphere_spherical_spring"]
for name in names:
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    x = g["x"]
    for dtype in ("f64", "mixed"):
        for opts in (None, {"loop_occ": 1}, {"loop_occ": 2}):
            sim = hip_backend.HipSim(m, x.shape[0], dtype=dtype, options=opts)
            xin = x.astype(np.float32).astype(np.float64) if dtype == "mixed" else x
            xt = torch.from_numpy(xin).to(sim.torch_dtype).cuda()

            def run():
                sim.x.copy_(xt)
                sim.step(None, 3)
                return sim.y.double().cpu().numpy()

            a = run()
            b = run()
            out = [f"{name} [{dtype}] {opts} n={x.shape[0]} out_dim={m.output_dim} nq={m.dof_q} nd={m.dof_qd}:"]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            out.append(f"clean/clean differ {int((~same).sum())}")
            for pattern in (0xFF, 0x7F):
                sim.debug_poison_lds(pattern)
                y = run()
                same = (y == a) | (np.isnan(y) & np.isnan(a))
                bad = np.argwhere(~same)
                cols = sorted(set(bad[:, 1].tolist()))
                envs = sorted(set(bad[:, 0].tolist()))
                nn = int(np.isnan(y).sum())
                md = float(np.nanmax(np.abs(y - a))) if bad.size else 0.0
                out.append(f"poison {pattern:#x}: differ {len(bad)} nan {nn} max|d| {md:.3e} cols {cols[:12]}{'...' if len(cols) > 12 else ''} "
                           f"envs {len(envs)} first {envs[:6]}")
            print("  ".join(out), flush=True)
            sim.close()
