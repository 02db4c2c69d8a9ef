#!/usr/bin/env python3
"""What a CONSUMER of the gathered records pays for the ring living in uncached device memory (hipDeviceMallocUncached:
peers store into it, tds_shard.hip peer_setup): a reduction over one gathered slot, a device-to-device copy of it and a
device-to-host copy into pinned memory, against an ordinary (cached) tensor of the same shape.  One rank holding as many
environments as an 8-rank run gathers per slot (8 x 4096 Ant records of 30 floats = 3.9 MB).
usage: python tools/experiments/uncached_ring_consumers.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tds_amd
from tds_amd import hip_backend

m = tds_amd.load_model("ant"); n = 8 * 4096
uid = hip_backend.HipShard.unique_id()
sh = hip_backend.HipShard(m, n, rank=0, world=1, device=0, dtype="f64", unique_id=uid, wire_dtype="f32", block=1)
sim = sh.sim
rng = np.random.default_rng(1)
x0 = np.zeros((n, m.input_dim)); ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
x0[:, 2] = 0.48; x0[:, 6:14] = ip + 0.05 * rng.uniform(-1, 1, (n, 8)); x0[:, -3:] = [15, 0.3, 3]
sim.x.copy_(torch.from_numpy(x0).cuda())
actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (4, n, 8))).cuda().contiguous()
sh.step_many(actions, 64)
sh.flush()
torch.cuda.synchronize()
g = sh.gathered()
print(f"exchange form {sh.exchange_form()}, gathered slot {tuple(g.shape)} {g.dtype}: {g.numel() * g.element_size() / 1e6:.2f} MB (uncached device memory)")
c = g.clone()  # (an ordinary allocation with the same contents)
host = torch.empty(g.shape, dtype=g.dtype).pin_memory()


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


out = torch.empty_like(c)
for name, fu, fc in (("reduction (sum over the slot)", lambda: g.sum(), lambda: c.sum()),
                     ("observation normalisation ((x - mean) * scale, elementwise, into a cached tensor)", lambda: torch.mul(g, 0.5, out=out), lambda: torch.mul(c, 0.5, out=out)),
                     ("device-to-device copy", lambda: out.copy_(g), lambda: out.copy_(c)),
                     ("device-to-host copy (pinned)", lambda: host.copy_(g, non_blocking=True), lambda: host.copy_(c, non_blocking=True))):
    tu, tc = timed(fu), timed(fc)
    mb = g.numel() * g.element_size() / 1e6
    print(f"{name}: uncached ring {tu:.1f} us ({mb / tu:.2f} TB/s), cached tensor {tc:.1f} us ({mb / tc:.2f} TB/s)")
