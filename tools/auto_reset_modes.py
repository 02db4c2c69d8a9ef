#!/usr/bin/env python3
"""Runs ON THE GPU BOX: VectorizedAntEnv.step with auto-reset — reset inside the step launch (step-loop build) vs
straight-line step + masked forced-reset launch (TDS_HIP_AUTO_RESET_SPLIT = 0 / 1 / unset = library's choice)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tds_amd  # noqa: E402

for n in (4096, 8192, 16384):
    for split in ("0", "1", None):
        if split is None:
            os.environ.pop("TDS_HIP_AUTO_RESET_SPLIT", None)
        else:
            os.environ["TDS_HIP_AUTO_RESET_SPLIT"] = split
        env = tds_amd.VectorizedAntEnv(n, auto_reset_when_done=True, seed=5)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = [(torch.rand((n, 8), dtype=torch.float64, device="cuda", generator=g) - 0.5) * 0.8 for _ in range(16)]
        for i in range(50):
            out = env.step(acts[i % 16])
        torch.cuda.synchronize()
        K, dones = 500, 0
        t0 = time.perf_counter()
        for i in range(K):
            out = env.step(acts[i % 16])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"ant x{n} auto-reset split={split}: {n * K / dt:.4g} env-steps/s (last step: {int(out.dones.sum())} done)", flush=True)
