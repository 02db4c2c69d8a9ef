#!/usr/bin/env python3
"""Phase stamps of the 8-lane kernel's step loop (csrc/tds_oct.hip built with -DTDS_OCT_PROF: tools/oct_profile.sh).
One 1000-step ring launch of Ant x N; workgroup 3 stamps the shader clock at the phase boundaries of iteration 500.
usage: python tools/oct_profile.py [n_envs=4096] [iter=500] [oct_w2=1]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TDS_HIP_LIB", os.path.join(ROOT, "tiny-differentiable-simulator_amd", "libtds_hip_octprof.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend

MAIN = ["PD, jcalc, root sincos, kinematics (-> barrier 1)", "D rigid inertias x 2", "E totals, G rows of M, H leg LDL^T + couplings",
        "Schur sums (LDS), root block R, 6 x 6 LDL^T", "F forward dynamics (-> barrier 2)", "waits for row windows + Gauss-Seidel sweep",
        "impulse, integrate, reward, reset pool (-> barrier 0)"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    it = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    m = tds_amd.load_model("ant")
    w2 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    sim = hip_backend.HipSim(m, n, options={"oct_w2": w2})
    assert sim.single_step_kernel()[0] == "oct8"
    rng = np.random.default_rng(3)
    x0 = np.zeros((n, m.input_dim))
    ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
    x0[:, 2] = 0.48
    x0[:, 6:14] = ip + 0.05 * rng.uniform(-1, 1, (n, 8))
    x0[:, -3:] = [15, 0.3, 3]
    sim.x.copy_(torch.from_numpy(x0).cuda())
    for _ in range(10):
        sim.step(None)
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, m.action_dim))).cuda().contiguous()
    slots = 64
    obs_ring = torch.zeros((slots, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
    y_ring = torch.zeros((slots, n, 160), dtype=torch.float64, device="cuda")
    L = hip_backend.lib()
    L.tds_oct_prof_read.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_ulonglong * 32)()
    assert L.tds_oct_prof_read(None, it) == 0
    for rep in range(3):
        sim.step_many_rings(actions, 1000, obs_ring, y_ring)
        torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    sim.step_many_rings(actions, 1000, obs_ring, y_ring)
    ev1.record()
    torch.cuda.synchronize()
    assert L.tds_oct_prof_read(buf, -1) == 0
    t = [int(buf[k]) for k in range(32)]
    print(f"ant x {n}: 1000-step ring launch {ev0.elapsed_time(ev1) * 1e3 / 1000:.2f} us per step (stamped build, option oct_w2 = {w2}); "
          f"iteration {it} of workgroup 3: NA = {t[15]}")
    print(f"  main wavefront, top of the step -> end of its step: {t[7] - t[0]} cycles")
    for k in range(7):
        print(f"  {t[k + 1] - t[k]:6d}  {MAIN[k]}")
    if w2:
        print(f"    of the contact phase: first window solved by the main wavefront {t[12] - t[5]}, swept {t[13] - t[12]}, remaining windows {t[6] - t[13]}")
    if w2:
        print("    rows of the first window, end of each row's chain after the window was solved: " + " ".join(str(t[16 + k] - t[12]) for k in range(8)))
    print("  helper (cycles relative to the main wavefront's top of step):")
    print(f"    narrowphase done at {t[8] - t[0]}, visual poses out at {t[9] - t[0]}, last row window solved at {t[10] - t[0]}, records out at {t[11] - t[0]}")


if __name__ == "__main__":
    main()
