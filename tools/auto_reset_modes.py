#!/usr/bin/env python3
"""Runs ON THE GPU BOX: VectorizedAntEnv.step with auto_reset_when_done under random actions (~5 % of the environments
end per step), per batch size and auto-reset form:
  split=0  reset + settle inside the step launch (step-loop build)
  split=1  straight-line step launch + forced-reset launch masked with the done flags
  split=2  reset pool: pre-settled states copied in by the straight-line kernel, refilled on a side stream
  None     the library's default (= the pool)
and the same loop with auto-reset OFF (the no-reset rate the pool is held against).  Then the same steps as calls of
step_many over 96 steps each (tds_hip_step_many with auto-reset on), with the random actions at two amplitudes
(+-0.4: ~5 % of the environments done per step; +-0.1: hardly any).   usage: tools/auto_reset_modes.py [many]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tds_amd  # noqa: E402


def run(n, auto, K=500):
    env = tds_amd.VectorizedAntEnv(n, auto_reset_when_done=auto, seed=5)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = [(torch.rand((n, 8), dtype=torch.float64, device="cuda", generator=g) - 0.5) * 0.8 for _ in range(16)]
    for i in range(50):
        out = env.step(acts[i % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        out = env.step(acts[i % 16])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n * K / dt, int(out.dones.sum())


def run_many(n, auto, amp=0.8, K=480, C=96):
    """the same loop as calls of step_many over C steps each (tds_hip_step_many with auto-reset on: step-loop launches
    that take the fresh states from the pool, where the plain call is one step-loop launch)"""
    env = tds_amd.VectorizedAntEnv(n, auto_reset_when_done=auto, seed=5)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = ((torch.rand((16, n, 8), dtype=torch.float64, device="cuda", generator=g) - 0.5) * amp).contiguous()
    out = env.step_many(acts, C)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K // C):
        out = env.step_many(acts, C)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n * (K // C) * C / dt, int(out.dones.sum()), env.sim.step_many_is_loop(C)


def many_rows(ns):
    for n in ns:
        for amp in (0.8, 0.2):
            base, _, loop = run_many(n, False, amp)
            v, d, loop_ar = run_many(n, True, amp)
            print(f"ant x{n} step_many (step-loop launches: {loop} without / {loop_ar} with auto-reset) actions +-{amp / 2}: no auto-reset {base:.4g}, auto-reset {v:.4g} "
                  f"env-steps/s = {v / base:.2f} x (last step: {d} done)", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "many":
    many_rows((4096, 8192))
    sys.exit(0)

for n in (4096, 8192, 16384):
    os.environ.pop("TDS_HIP_AUTO_RESET_SPLIT", None)
    base, _ = run(n, False)
    print(f"ant x{n} no auto-reset: {base:.4g} env-steps/s", flush=True)
    for split in ("0", "1", "2", None):
        if split is None:
            os.environ.pop("TDS_HIP_AUTO_RESET_SPLIT", None)
        else:
            os.environ["TDS_HIP_AUTO_RESET_SPLIT"] = split
        v, d = run(n, True)
        print(f"ant x{n} auto-reset split={split}: {v:.4g} env-steps/s = {v / base:.2f} x no-reset (last step: {d} done)", flush=True)
many_rows((4096, 8192, 16384))
