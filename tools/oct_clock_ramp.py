#!/usr/bin/env python3
"""Shader-clock frequency over the first 32 steps of a step-loop launch of the 8-lane kernel (profiling build of the library:
tools/oct_profile.sh).  The main wavefront of one workgroup reads the shader clock (s_memtime) and the 100 MHz real-time clock
(s_memrealtime) at the top of every step: cycles per step, microseconds per step and their ratio — what a SHORT launch (the
driver's 20 steps) pays at its beginning compared with the steady state.
usage: python tools/oct_clock_ramp.py [n_envs=4096] [idle_ms_before_launch=0]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TDS_HIP_LIB", os.path.join(ROOT, "tiny-differentiable-simulator_amd", "libtds_hip_octprof.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    m = tds_amd.load_model("ant")
    sim = hip_backend.HipSim(m, n)
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
    obs_ring = torch.zeros((64, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
    y_ring = torch.zeros((64, n, 160), dtype=torch.float64, device="cuda")
    L = hip_backend.lib()
    buf = (C.c_ulonglong * 64)()
    for K in (20, 32):
        for rep in range(3):  # (warm: code, rings)
            sim.step_many_rings(actions, 1000, obs_ring, y_ring)
        torch.cuda.synchronize()
        if idle_ms > 0:
            time.sleep(idle_ms * 1e-3)
        sim.step_many_rings(actions, K, obs_ring, y_ring)
        torch.cuda.synchronize()
        assert L.tds_oct_prof_clocks(buf) == 0
        nwg = (n + 7) // 8
        wb = (C.c_ulonglong * (3 * nwg))()
        assert L.tds_oct_prof_workgroups(wb, nwg) == 0
        st = np.array([int(wb[3 * i]) for i in range(nwg)], dtype=np.int64)
        lo = np.array([int(wb[3 * i + 1]) for i in range(nwg)], dtype=np.int64)
        en = np.array([int(wb[3 * i + 2]) for i in range(nwg)], dtype=np.int64)
        t0 = st.min()
        q = lambda a: " / ".join("%.1f" % (v / 100.0) for v in np.percentile(a - t0, [0, 10, 50, 90, 100]))
        print(f"ant x {n}: {nwg} workgroups of a {K}-step launch, us after the first workgroup's first instruction (min / 10 % / median / 90 % / max):")
        print(f"   first instruction {q(st)}   top of the first step {q(lo)}   behind the last step's stores {q(en)}")
        print(f"   per workgroup: prologue (table, records) median {np.median(lo - st) / 100.0:.1f} us, its {K} steps + epilogue median {np.median(en - lo) / 100.0:.1f} us; "
              f"the launch as the workgroups see it: {(en.max() - t0) / 100.0:.1f} us")
        c = [int(buf[2 * k]) for k in range(K)]
        r = [int(buf[2 * k + 1]) for k in range(K)]
        print(f"ant x {n}: a {K}-step launch {idle_ms} ms after a 1000-step launch (stamped build); per step: shader cycles | us (100 MHz clock) | MHz")
        for k in range(K - 1 if K == 20 else 0):
            dc, dr = c[k + 1] - c[k], (r[k + 1] - r[k]) / 100.0
            print(f"  step {k:2d}: {dc:7d} cycles  {dr:7.2f} us  {dc / dr if dr > 0 else 0:7.0f} MHz")


main()
