import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tds_amd
from tds_amd import hip_backend
m = tds_amd.load_model("laikago")
g = np.load(os.path.join("tests", "golden", "laikago.npz"))
n = 4
x = np.tile(g["x"][3], (n, 1))
x[0, 2] -= float(sys.argv[2]) ; x[0, 3] += 0.15           # env 0: tilted: some toes down
x[1:, 2] += 5.0                              # mates: airborne
sim = hip_backend.HipSim(m, n, dtype="f64")
y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
np.save(sys.argv[1], y[0, 36:36 + 14 * 12 + 23])
print("na of env 0:", y[0, 36 + 14 * 12 + 22])
