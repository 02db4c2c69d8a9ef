"""Timing-dependent defects of the two-wavefront kernels (LDS flags instead of barriers): the same step many times,
every result compared bit for bit with the first one; a concurrent stream keeps the GPU busy half of the time."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, tds_amd
from tds_amd import hip_backend
from conftest import GOLDEN
names = sys.argv[1:] or ["cartpole_plane", "pendulum5_plane", "ant", "laikago"]
side = torch.cuda.Stream()
junk = torch.randn(4096, 4096, device="cuda")
for name in names:
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    for dtype in ("mixed", "f64"):
        for reps_n in (1, 64):  # the golden batch, and it tiled (more workgroups in flight)
            x = np.tile(g["x"], (reps_n, 1))
            sim = hip_backend.HipSim(m, x.shape[0], dtype=dtype)
            xt = torch.from_numpy(x).to(sim.torch_dtype).cuda()
            ref = sim.forward_zero(xt).clone()
            torch.cuda.synchronize()
            bad = 0
            for it in range(400):
                if it % 2:
                    with torch.cuda.stream(side):
                        junk @ junk
                y = sim.forward_zero(xt)
                if not torch.equal(y.view(torch.int32 if dtype == "mixed" else torch.int64), ref.view(torch.int32 if dtype == "mixed" else torch.int64)):
                    bad += 1
            torch.cuda.synchronize()
            print(f"{name} {dtype} x{x.shape[0]}: {bad} of 400 repetitions differ", flush=True)
