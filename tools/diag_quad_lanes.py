import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tds_amd
from tds_amd import hip_backend
m = tds_amd.load_model("laikago")
g = np.load(os.path.join("tests", "golden", "laikago.npz"))
n = min(64, g["x"].shape[0])
x = g["x"][:n].copy()
x[:, 2] -= 0.1
sim = hip_backend.HipSim(m, n, dtype="f64")
y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
d = y[0, 36:36 + 16 * 23].reshape(16, 23)
names = ["ids"] * 6 + ["qdr"] * 6 + ["Cr"] * 6 + ["It"] * 3 + ["ft"] * 2
for k in range(23):
    col = d[:, k]
    if not np.all(col == col[0]):
        print(names[k], k, "differs across lanes:", ["%.17g" % v for v in col[[0, 1, 4, 5, 8, 12]]])
print("done")
