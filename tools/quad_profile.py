#!/usr/bin/env python3
"""Phase stamps of the 16-lane kernel (csrc/tds_quad.hip built with -DTDS_QUAD_PROF: tools/quad_profile.sh).
Workgroup 3 stamps the shader clock at the phase boundaries of one step: of iteration `iter` of a 1000-step ring launch
(the step-loop form), or of a plain single-step launch (the straight-line form: what laikago_soft x 8192 runs).
usage: python tools/quad_profile.py [n_envs=8192] [form=single|loop] [iter=500]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TDS_HIP_LIB", os.path.join(ROOT, "tiny-differentiable-simulator_amd", "libtds_hip_quadprof.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend

PH = ["PD", "B jcalc (own sincos; the toes' lanes: the root's)", "C root chain in closed form", "leg scans (pose, velocity, bias acceleration)",
      "D rigid inertia + bias force of my link", "I narrowphase (toes) + contact count", "M1 visual poses of y, root body's rigid inertia",
      "E composites (CRBA suffix sums), root axes, C of the root", "G rows of M", "H LDL^T: leg blocks, couplings, Schur complement, 6 x 6",
      "F forward dynamics + integrate_euler_qdd", "J K L contact rows + Gauss-Seidel sweep + impulse", "M integrate_euler, state -> record",
      "y record stores", "N reward / done", "reset pool", "obs record / state stores"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    form = sys.argv[2] if len(sys.argv) > 2 else "single"
    it = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    m = tds_amd.load_model("laikago_soft")
    sim = hip_backend.HipSim(m, n, options={"step_many_loop": 1} if form == "loop" else None)
    assert sim.single_step_kernel()[0] == "quad16"
    rng = np.random.default_rng(3)
    nq, adim = m.dof_q, m.action_dim
    x0 = np.zeros((n, m.input_dim))
    ip = np.array([m.initial_poses[i] for i in range(adim)])
    x0[:, 2] = 0.48
    x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
    x0[:, -3:] = [100, 2, 50]
    sim.x.copy_(torch.from_numpy(x0).cuda())
    for _ in range(10):
        sim.step(None)
    actions = torch.from_numpy(rng.uniform(-0.1, 0.1, (16, n, adim))).cuda().contiguous()
    L = hip_backend.lib()
    L.tds_quad_prof_read.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 32)()
    assert L.tds_quad_prof_read(None, it) == 0
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if form == "loop":
        slots = 64
        obs_ring = torch.zeros((slots, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
        y_ring = torch.zeros((slots, n, 416), dtype=torch.float64, device="cuda")
        assert sim.step_many_is_loop(1000)
        for _ in range(2):
            sim.step_many_rings(actions, 200, obs_ring, y_ring)
        torch.cuda.synchronize()
        ev0.record()
        sim.step_many_rings(actions, 1000, obs_ring, y_ring)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / 1000
        what = f"1000-step ring launch, iteration {it}"
    else:
        obs = torch.zeros((n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
        for k in range(200):
            sim.step(actions[k % 16], 1, obs)
        torch.cuda.synchronize()
        ev0.record()
        for k in range(100):
            sim.step(actions[k % 16], 1, obs)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / 100
        what = "single-step launches (eager), the last launch"
    assert L.tds_quad_prof_read(buf, -1) == 0
    t = [int(buf[k]) for k in range(32)]
    print(f"laikago_soft x {n}, {form} form: {us:.2f} us per step (stamped build); {what}, workgroup 3: NA = {t[18]}")
    print(f"  top of the step -> end of the step: {t[17] - t[0]} cycles (shader clock, 100 MHz s_memtime ticks scaled by the hardware)")
    for k in range(17):
        print(f"  {t[k + 1] - t[k]:6d}  {PH[k]}")


if __name__ == "__main__":
    main()
