"""Hunt for reads of uninitialised device / LDS memory: fresh handles between allocations poisoned with NaN / huge values,
one forward step each, against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, tds_amd, oraclelib
from tds_amd import hip_backend
from conftest import rel_err, GOLDEN
names = sys.argv[1:] or ["cartpole_plane", "pendulum5_plane", "ant", "cartpole"]
bad = 0
for name in names:
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    for dtype in ("mixed", "f64"):
        x = g["x"].astype(np.float32).astype(np.float64) if dtype == "mixed" else g["x"]
        yr = oraclelib.step(m, x)
        worst = 0.0
        for it in range(60):
            # poison: allocate, fill, free — the next hipMalloc of the handle tends to land on it
            junk = [torch.full((1 << 20,), float("nan") if it % 2 else 1e300, dtype=torch.float64, device="cuda") for _ in range(8)]
            torch.cuda.synchronize()
            del junk
            torch.cuda.empty_cache()
            sim = hip_backend.HipSim(m, x.shape[0], dtype=dtype)
            xt = torch.from_numpy(x).to(sim.torch_dtype).cuda()
            y = sim.forward_zero(xt).double().cpu().numpy()
            e = rel_err(y, yr)
            worst = max(worst, e if np.isfinite(e) else 1e30)
            if not (e < 1e-6):
                bad += 1
                rows = np.nonzero(~(np.abs(y - yr) <= 1e-5 * (1 + np.abs(yr))).all(axis=1))[0]
                print(f"  {name} {dtype} iteration {it}: err {e:.3e}, rows {rows[:10]}", flush=True)
            del sim
        print(f"{name} {dtype}: worst {worst:.3e}", flush=True)
print("bad:", bad)
