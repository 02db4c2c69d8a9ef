import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, torch
import tds_amd
from tds_amd import hip_backend
import oraclelib
import test_chain as tc
from conftest import rel_err
import ctypes as C

def variant(n_links, seed, keep):
    full = tc._synthetic_chain(n_links, seed)
    base = tds_amd.load_model("pendulum5")
    m = tc._synthetic_chain(n_links, seed)
    for i in range(n_links):
        l, b = m.links[i], base.links[min(i, 4)]
        if "joint" not in keep:
            l.joint_type = b.joint_type
            for k in range(6): l.S[k] = b.S[k]
        if "xt" not in keep:
            for k in range(9): l.X_T_rot[k] = b.X_T_rot[k]
            for k in range(3): l.X_T_trans[k] = b.X_T_trans[k]
        if "inertia" not in keep:
            l.mass = b.mass
            for k in range(3): l.com[k] = b.com[k]
            for k in range(9): l.inertia[k] = b.inertia[k]
        if "spring" not in keep:
            l.stiffness = 0.0; l.damping = 0.0
        if "vis" not in keep:
            bv = base.visuals[min(i, 4)]
            for k in range(9): m.visuals[i].X_rot[k] = bv.X_rot[k]
            for k in range(3): m.visuals[i].X_trans[k] = bv.X_trans[k]
    if "base" not in keep:
        for k in range(9): m.base_X_world_rot[k] = base.base_X_world_rot[k]
        for k in range(3): m.base_X_world_trans[k] = base.base_X_world_trans[k]
    return m

n_links, seed = int(sys.argv[1]), int(sys.argv[2])
for keep in [(), ("joint",), ("xt",), ("inertia",), ("spring",), ("vis",), ("base",), ("joint","xt"), ("joint","xt","inertia","spring","vis","base")]:
    m = variant(n_links, seed, keep)
    rng = np.random.default_rng(7)
    n = 16
    x = np.zeros((n, m.input_dim))
    x[:, :n_links] = rng.uniform(-2.5, 2.5, (n, n_links))
    x[:, n_links:2*n_links] = rng.uniform(-2, 2, (n, n_links))
    x[:, 2*n_links:] = rng.uniform(-3, 3, (n, n_links))
    y_ref = oraclelib.step(m, x)
    out = []
    for opts in (None, {"chain": 0}):
        sim = hip_backend.HipSim(m, n, options=opts)
        y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
        out.append((sim.single_step_kernel()[0], rel_err(y[:, :n_links], y_ref[:, :n_links]), rel_err(y[:, n_links:2*n_links], y_ref[:, n_links:2*n_links]), rel_err(y[:, 2*n_links:], y_ref[:, 2*n_links:])))
    print(keep, " ".join("%s q %.1e qd %.1e vis %.1e |" % o for o in out))
