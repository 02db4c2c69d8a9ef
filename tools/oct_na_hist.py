#!/usr/bin/env python3
"""How many contact points penetrate per Ant environment, and the largest count per group of 8 consecutive environments (what the
8-lane kernel's sweep is laid out for), along the bench workload: 4096 environments, +-0.4 actions.  From the y records of a ring
launch: the visual poses ARE the links' world transforms (locomotion_contact_simulation.h:281-299).
usage: python tools/oct_na_hist.py [n_envs=4096] [steps=200]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend


def quat_rot(q, v):  # q = (x, y, z, w) per row, v a 3-vector
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n
    R = np.stack([1 - s * (y * y + z * z), s * (x * y - w * z), s * (x * z + w * y),
                  s * (x * y + w * z), 1 - s * (x * x + z * z), s * (y * z - w * x),
                  s * (x * z - w * y), s * (y * z + w * x), 1 - s * (x * x + y * y)], -1).reshape(q.shape[:-1] + (3, 3))
    return R @ v


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
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
    obs_ring = torch.zeros((steps, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
    y_ring = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
    for rep in range(6):  # 6 x steps: the statistics of the last launch (well into the random-action regime)
        sim.step_many_rings(actions, steps, obs_ring, y_ring)
    y = y_ring.cpu().numpy()
    # visual k of link 5 + k: pose = X_world * X_visual; the capsule ends / the sphere centre in the VISUAL frame are +-L/2 e_z / 0
    # (the Ant's collision and visual frames coincide: models/ant.json)
    cnt = np.zeros((steps, n), int)
    for k in range(9):
        pose = y[:, :, 28 + 7 * k:28 + 7 * k + 7]
        pos, quat = pose[..., :3], pose[..., 3:7]
        g = m.geoms[k]
        if k == 0:
            cnt += (pos[..., 2] - g.radius < 0)
        else:
            for sgn in (0.5, -0.5):
                c = pos + quat_rot(quat, np.array([0.0, 0.0, sgn * g.length]))
                cnt += (c[..., 2] - g.radius < 0)
    per_env = np.bincount(cnt.ravel(), minlength=18) / cnt.size
    wave = cnt.reshape(steps, n // 8, 8).max(-1)
    per_wave = np.bincount(wave.ravel(), minlength=18) / wave.size
    grp4 = cnt.reshape(steps, n // 4, 4).max(-1)
    per_g4 = np.bincount(grp4.ravel(), minlength=18) / grp4.size
    print(f"ant x {n}, {steps} steps after {5 * steps}: contacts per environment mean {cnt.mean():.2f}; per group of 8 (max) mean {wave.mean():.2f}, "
          f"windows of 8 rows mean {np.ceil(3 * wave / 8).mean():.2f}; per group of 4 (max) mean {grp4.mean():.2f}")
    print("count  per-env  per-8-group  per-4-group")
    for c in range(18):
        if per_env[c] + per_wave[c] + per_g4[c] > 0:
            print(f"{c:5d}  {per_env[c]:7.4f}  {per_wave[c]:7.4f}  {per_g4[c]:7.4f}")


if __name__ == "__main__":
    main()
