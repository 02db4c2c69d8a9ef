import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tds_amd
from tds_amd import hip_backend
name = "laikago"
m = tds_amd.load_model(name)
g = np.load(os.path.join("tests", "golden", name + ".npz"))
n = 8192
rng = np.random.default_rng(11)
idx = rng.integers(0, g["x"].shape[0], n)
x = g["x"][idx] + 1e-3 * rng.standard_normal((n, m.input_dim)) * (np.arange(m.input_dim) < m.dof_q + m.dof_qd)
sim = hip_backend.HipSim(m, n, dtype="f64")
xd = torch.from_numpy(x).cuda()
y = sim.forward_zero(xd).clone()
perm = torch.randperm(n, device="cuda")
y2 = sim.forward_zero(xd[perm].contiguous()).clone()
d = (y2 != y[perm])
bad = d.any(dim=1).nonzero().flatten()
orig = perm[bad].tolist()
print("differing (original indices):", orig[:12])
# toe heights -> contact counts per env: use the oracle-free proxy: run general kernel? simpler: count via a second quantity
# experiment: env e0 with different wavefront mates
e0 = orig[0]
air = x[e0].copy(); air[2] += 5.0            # far above the plane
low = x[e0].copy(); low[2] -= 0.2            # pushed into the plane: all four toes down
def run(rows):
    b = np.tile(x[e0], (n, 1))
    for i, r in enumerate(rows):
        b[i] = r
    return sim.forward_zero(torch.from_numpy(b).cuda()).clone()
base = run([x[e0]] * 4)[0]
for label, rows, pos in (("alone x4", [x[e0]] * 4, 0), ("with 3 airborne", [x[e0], air, air, air], 0), ("with 3 all-down", [x[e0], low, low, low], 0),
                         ("position 1, 3 all-down", [low, x[e0], low, low], 1), ("position 3, airborne", [air, air, air, x[e0]], 3),
                         ("position 2 alone-like", [x[e0]] * 4, 2)):
    out = run(rows)[pos]
    nd = int((out != base).sum())
    print(f"{label:28s}: {nd} differing scalars vs 'alone x4'", ("max abs %.3e" % float((out - base).abs().max())) if nd else "")
