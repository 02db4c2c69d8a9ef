#!/usr/bin/env python3
"""One rank through the ring exchange (tds_hip_shard_step_many), nothing else: for rocprofv3 --kernel-trace.
usage: python tools/trace_ring_exchange.py [rccl=1] [n=4096] [chunks=4] [key=value ...]   (library options of the shard)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import tds_amd
from tds_amd import hip_backend

rccl = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
m = tds_amd.load_model("ant")
rng = np.random.default_rng(3)
x0 = np.zeros((n, m.input_dim))
ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
x0[:, 2] = 0.48
x0[:, 6:m.dof_q] = ip + 0.05 * rng.uniform(-1, 1, (n, m.dof_q - 6))
x0[:, -3:] = [15, 0.3, 3]
uid = hip_backend.HipShard.unique_id() if rccl else None
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[4:] if "=" in a}
sh = hip_backend.HipShard(m, n, unique_id=uid, wire_dtype="f32", options=opts)
sh.sim.x.copy_(torch.from_numpy(x0).cuda())
acts = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, m.action_dim))).cuda().contiguous()
for _ in range(chunks):
    sh.step_many(acts, 64)
sh.flush()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(chunks):
    sh.step_many(acts, 64)
sh.flush()
torch.cuda.synchronize()
print("us per step:", (time.perf_counter() - t0) / (64 * chunks) * 1e6)
