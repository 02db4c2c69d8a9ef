"""Long closed loops of the two benchmark robots against the REAL reference (oracle/_ref/libtds_ref.so on the host cores),
per-step resync, every environment: the states the kernels' re-associated kinematics (closed-form root chain, leg scan) and
factorisation meet over ~1000 steps — fallen robots, many contacts — not just the first 100.
usage: python tools/long_closed_loop_vs_reference.py [steps]   (diagnostic; tests/test_hip_parity.py holds the gated 100-step form)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import tds_amd
from tds_amd import hip_backend
from test_hip_parity import _reference_stepper

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for name, n, amp, var in (("ant", 4096, 0.4, [15, 0.3, 3]), ("laikago_soft", 2048, 0.1, [100, 2, 50]), ("pendulum5", 4096, 1.0, None)):
    m = tds_amd.load_model(name)
    ref_step, what = _reference_stepper(name, n)
    rng = np.random.default_rng(2025)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    x0 = np.zeros((n, m.input_dim))
    if var is not None:
        ip = np.array([m.initial_poses[i] for i in range(adim)])
        x0[:, 2] = 0.48
        x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
        x0[:, -3:] = var
    else:  # (torque-driven chain: random joint angles and velocities)
        x0[:, :nq] = rng.uniform(-1, 1, (n, nq))
        x0[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
    sim = hip_backend.HipSim(m, n)
    sim.x.copy_(torch.from_numpy(x0).cuda())
    for _ in range(10 if var is not None else 0):
        sim.step(None)
    worst, worst_t, wild, hist = 0.0, -1, 0, []
    for t in range(steps):
        a = rng.uniform(-amp, amp, (n, adim))
        x_before = sim.x.cpu().numpy()
        x_before[:, nq + nd:nq + nd + adim] = a
        sim.step(torch.from_numpy(a).cuda())
        y_ref = ref_step(x_before)
        y = sim.y.cpu().numpy()
        calm = np.isfinite(y_ref).all(axis=1) & (np.abs(y_ref[:, nq:nq + nd]).max(axis=1) < 1e3)
        wild += int((~calm).sum())
        e = float(np.max(np.abs(y[calm] - y_ref[calm]) / np.maximum(np.abs(y_ref[calm]), 1e-3)))
        if e > worst:
            worst, worst_t = e, t
        if (t + 1) % 200 == 0:
            up = x_before[:, 2]
            hist.append(f"step {t + 1}: worst so far {worst:.2e}" + (f", torso z min / median {up.min():.2f} / {np.median(up):.2f}" if var is not None else ""))
    print(f"{name} x{n} [{sim.single_step_kernel()[0]}], {steps} closed-loop steps (actions +-{amp}), every environment vs {what}: worst per-step rel err "
          f"{worst:.3e} (step {worst_t}); environment-steps beyond |qd| = 1e3 or non-finite in the reference (excluded): {wild}")
    for h in hist:
        print("   ", h)
