"""One cold first launch per process (code objects not loaded, caches cold): result against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, tds_amd, oraclelib
from tds_amd import hip_backend
from conftest import rel_err, GOLDEN
name, dtype = sys.argv[1], sys.argv[2]
m = tds_amd.load_model(name)
g = np.load(os.path.join(GOLDEN, name + ".npz"))
x = g["x"].astype(np.float32).astype(np.float64) if dtype == "mixed" else g["x"]
sim = hip_backend.HipSim(m, x.shape[0], dtype=dtype)
y = sim.forward_zero(torch.from_numpy(x).to(sim.torch_dtype).cuda()).double().cpu().numpy()
e = rel_err(y, oraclelib.step(m, x))
y2 = sim.forward_zero(torch.from_numpy(x).to(sim.torch_dtype).cuda()).double().cpu().numpy()
print(name, dtype, "cold err %.3e" % e, "warm equals cold:", np.array_equal(y, y2), "BAD" if not e < 1e-6 else "")
