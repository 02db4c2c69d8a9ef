"""Long closed loops of the multi-body worlds against the C oracle with per-step resync (diagnostic): worlds whose chains
interpenetrate under constant torques blow up numerically in the reference's algorithm too — are the HIP kernels and the
oracle together until then?  Per step: environments that are finite and calm (|qd| < 1e3) in the oracle are held to 1e-6."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import tds_amd, oraclelib
from tds_amd import hip_backend
import gen_golden as gen

n, T = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 250
for name in ["two_pendulums_plane", "three_pendulums", "four_pendulums", "two_cubes_floating"]:
    m = tds_amd.load_model(name)
    nq, nd = m.dof_q, m.dof_qd
    x = gen.random_inputs(name, m, n, np.random.default_rng(5))
    sim = hip_backend.HipSim(m, n)
    worst, blown, restarted = 0.0, 0, 0
    for t in range(T):
        y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
        y_ref = oraclelib.step(m, x, threads=32)
        ok = np.isfinite(y_ref).all(axis=1) & (np.abs(y_ref[:, nq:nq + nd]).max(axis=1) < 1e3)
        if ok.any():
            e = float(np.max(np.abs(y[ok] - y_ref[ok]) / np.maximum(np.abs(y_ref[ok]), 1e-3)))
            worst = max(worst, e)
            assert e < 1e-6, (name, t, e)
        # environments the oracle itself has lost: both sides non-finite or huge together?
        lost = ~ok
        blown += int(lost.sum())
        x[:, :nq + nd] = y_ref[:, :nq + nd]
        if lost.any():
            fresh = gen.random_inputs(name, m, int(lost.sum()), np.random.default_rng(1000 + t))
            x[lost] = fresh
            restarted += int(lost.sum())
    print(f"{name}: {n} worlds x {T} closed-loop steps, worst per-step rel err vs the oracle on calm worlds {worst:.2e}; "
          f"{restarted} worlds blew up in the ORACLE along the way (|qd| > 1e3 or non-finite) and were restarted")
