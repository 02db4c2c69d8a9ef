"""env-steps/s of the multi-body worlds (kernel kinds 3 and 4; coverage, not tuned): 4096 worlds, closed loop on device,
one straight-line launch per step and K steps per step-loop launch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import tds_amd
from tds_amd import hip_backend
import gen_golden as gen

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name in ["two_pendulums_plane", "three_pendulums", "three_pendulums_plane", "four_pendulums", "two_cubes_floating",
             "pendulum_and_cube"]:
    m = tds_amd.load_model(name)
    x = gen.random_inputs(name, m, n, np.random.default_rng(5))
    sim = hip_backend.HipSim(m, n)
    sim.x.copy_(torch.from_numpy(x).cuda())
    a = torch.from_numpy(x[:, m.dof_q + m.dof_qd:m.dof_q + m.dof_qd + m.action_dim].copy()).cuda().contiguous() \
        if m.action_dim else None
    for _ in range(20):
        sim.step(a, 1)
    torch.cuda.synchronize()
    res = []
    for nsub, reps in ((1, 200), (50, 8)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            sim.step(a, nsub)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * nsub)
        res.append(f"{nsub} step(s) per launch: {us:.1f} us per step = {n / us * 1e6:.3g} env-steps/s")
    info = sim.kernel_info() if hasattr(sim, "kernel_info") else {}
    finite = bool(torch.isfinite(sim.x).all())
    print(f"{name} x{n} ({m.num_bodies} bodies, {m.dof_qd} dof, lanes/env {info.get('lanes_per_env', '?')}): " + "; ".join(res) +
          f"; state finite: {finite}")
