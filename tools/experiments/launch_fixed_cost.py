#!/usr/bin/env python3
"""Where the fixed cost of a SHORT step-loop launch goes (the driver's line is 20 steps per launch): kernel time from a HIP
event pair around K-step launches, K = 1 ... 320, for the record forms a launch can have.
usage: python tools/experiments/launch_fixed_cost.py [model=ant] [n=4096]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "ant"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    m = tds_amd.load_model(name)
    rng = np.random.default_rng(3)
    x0 = np.zeros((n, m.input_dim))
    ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
    x0[:, 2] = 0.48
    x0[:, 6:m.dof_q] = ip + 0.05 * rng.uniform(-1, 1, (n, m.dof_q - 6))
    x0[:, -3:] = [15, 0.3, 3] if name.startswith("ant") else [100, 2, 50]
    actions = torch.from_numpy(rng.uniform(-0.1, 0.1, (16, n, m.action_dim))).cuda().contiguous()
    ystr = -(-m.output_dim // 16) * 16

    def handle(opts=None):
        s = hip_backend.HipSim(m, n, options=opts)
        s.x.copy_(torch.from_numpy(x0).cuda())
        for _ in range(10):
            s.step(None)
        return s

    def t_of(fn, reps=15):
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        return best

    Ks = [1, 2, 5, 10, 20, 40, 80, 160, 320]
    forms = {}
    s = handle()
    obs = torch.zeros((n, s.obs_dim + 2), dtype=torch.float64, device="cuda")
    forms["no rings (records of the last step)"] = lambda K: t_of(lambda: s.step_many(actions, K, obs))
    for RS in (64, 2):
        obs_ring = torch.zeros((RS, n, s.obs_dim + 2), dtype=torch.float64, device="cuda")
        y_ring = torch.zeros((RS, n, ystr), dtype=torch.float64, device="cuda")
        forms[f"rings of {RS} slots (y on 128-byte lines)"] = (
            lambda K, o=obs_ring, y=y_ring: t_of(lambda: s.step_many_rings(actions, K, o, y)))
        forms[f"obs ring only, {RS} slots"] = (lambda K, o=obs_ring: t_of(lambda: s.step_many_rings(actions, K, o, None)))
    print(f"{name} x {n}: kernel us of ONE K-step launch (min of 15, HIP events), and (us - K * us_per_step_at_320)")
    print("%-46s" % "form" + "".join("%9d" % K for K in Ks))
    for what, f in forms.items():
        ts = [f(K) for K in Ks]
        per = ts[-1] / Ks[-1]
        print("%-46s" % what + "".join("%9.1f" % t for t in ts))
        print("%-46s" % ("   fixed part (per step %.2f us)" % per) + "".join("%9.1f" % (t - K * per) for t, K in zip(ts, Ks)))


main()
