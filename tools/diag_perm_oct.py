#!/usr/bin/env python3
"""Does an Ant environment's result depend on its wavefront-mates (their contact counts decide the sweep's layout) or on its
position in the batch?  Whole batch vs a permutation vs ragged sub-batches, bit for bit; states with 0 .. 17 contacts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import tds_amd
from tds_amd import hip_backend
from test_oct import _contact_states
m = tds_amd.load_model("ant")
n = 4096
x = _contact_states(m, n, np.random.default_rng(11))
for w2 in (1, 0):
    sim = hip_backend.HipSim(m, n, options={"oct_w2": w2})
    xd = torch.from_numpy(x).cuda()
    y = sim.forward_zero(xd).clone()
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    y2 = sim.forward_zero(xd[perm].contiguous())
    d = (y2.view(torch.int64) != y[perm].view(torch.int64))
    bad_env = d.any(1).nonzero().flatten().cpu().numpy()
    print("oct_w2", w2, ": permuted batch, mismatching envs", len(bad_env), "of", n)
    # ragged sub-batches: 171 environments at a time through a handle of 171
    sub = hip_backend.HipSim(m, 171, options={"oct_w2": w2})
    bad = 0
    for e0 in range(0, 171 * 5, 171):
        ys = sub.forward_zero(xd[e0:e0 + 171].contiguous())
        bad += int((ys.view(torch.int64) != y[e0:e0 + 171].view(torch.int64)).any(1).sum())
    print("oct_w2", w2, ": sub-batches of 171, mismatching envs", bad, "of", 171 * 5)
    if len(bad_env):
        cols = d.any(0).nonzero().flatten().cpu().numpy()
        print(" columns that differ:", cols[:40])
        yy = y[perm].cpu().numpy(); y2n = y2.cpu().numpy()
        pe = perm.cpu().numpy()
        print(" first bad envs: new index -> old index:", [(int(b), int(pe[b])) for b in bad_env[:8]])
        e = bad_env[0]
        print(" env", e, "rel diff", (np.abs(yy[e]-y2n[e])/np.maximum(np.abs(yy[e]),1e-3)).max())
