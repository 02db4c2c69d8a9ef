"""Where the fixed cost of a 20-step call goes (Ant x 4096, per-step records): host time until the call returns, until the
GPU is done (synchronise), for the plain ring call (tds_hip_step_many_rings) and for one rank through the shard layer
(tds_hip_shard_step_many + tds_hip_shard_flush), peer-store form with 0 / 7 loopback peers and the RCCL group form.
usage: python tools/call_overhead.py [K]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tds_amd
from tds_amd import hip_backend

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = tds_amd.load_model("ant")
n = 4096
rng = np.random.default_rng(3)
nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
x0 = np.zeros((n, m.input_dim)); x0[:, 2] = 0.48
x0[:, 6:nq] = np.array([m.initial_poses[i] for i in range(adim)]) + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
x0[:, -3:] = [15, 0.3, 3]
a = torch.from_numpy(rng.uniform(-0.4, 0.4, (16, n, adim))).cuda().contiguous()
RS = 64
ALT = int(os.environ.get("ALT", "0"))  # experiment slot of the library (tools/build_alt.sh), 0: its own kernels


def measure(call, wait, label, reps=40):
    for _ in range(5):
        call(); wait(); torch.cuda.synchronize()
    t_ret, t_wait, t_all = [], [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        t1 = time.perf_counter()
        wait()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        t_ret.append(t1 - t0); t_wait.append(t2 - t1); t_all.append(t3 - t0)
    med = lambda v: 1e6 * float(np.median(v))
    print(f"{label:58s} call returns {med(t_ret):6.1f} us | flush {med(t_wait):6.1f} us | GPU done {med(t_all):6.1f} us "
          f"= {med(t_all) / K:5.2f} us/step")
    return med(t_all)


sim = hip_backend.HipSim(m, n, options={"alt_build": ALT} if ALT else None)
sim.x.copy_(torch.from_numpy(x0).cuda())
obs_ring = torch.zeros((RS, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
y_ring = torch.zeros((RS, n, 160), dtype=torch.float64, device="cuda")
base = measure(lambda: sim.step_many_rings(a, K, obs_ring, y_ring), lambda: None, f"plain ring call, {K} steps")
long = measure(lambda: sim.step_many_rings(a, 1000, obs_ring, y_ring), lambda: None, "plain ring call, 1000 steps", reps=5) / 1000 * K
print(f"   {K} steps at the 1000-step rate: {long:.1f} us -> fixed cost of a plain call {base - long:.1f} us")
for opts, label in ((None, "shard, peer stores, 0 peers"), ({"shard_peer_loopback": 7}, "shard, peer stores, 7 loopback peers"),
                    ({"shard_peer": 0}, "shard, RCCL group behind the launch")):
    uid = hip_backend.HipShard.unique_id()
    if ALT:
        opts = dict(opts or {}, alt_build=ALT)
    sh = hip_backend.HipShard(m, n, rank=0, world=1, device=0, dtype="f64", unique_id=uid, wire_dtype="f32", options=opts)
    sh.sim.x.copy_(torch.from_numpy(x0).cuda())
    t = measure(lambda: sh.step_many(a, K), lambda: sh.flush(), f"{label} ({sh.exchange_form() or '-'}), {K} steps")
    print(f"   over the plain call: {t - base:+.1f} us")
    sh.close()
