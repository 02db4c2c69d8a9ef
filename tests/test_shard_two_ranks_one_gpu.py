"""The C shard layer (tds_hip_shard_*, csrc/tds_shard.hip) with TWO RANKS — two processes on the ONE GPU of the test box.

The real RCCL refuses two ranks on one device, so the exchange goes through tests/stub_rccl (a stream-ordered
all-gather over POSIX shared memory, loaded through TDS_HIP_RCCL_LIB: the shard layer resolves every nccl* symbol with
dlsym).  Everything else is the product path: rendezvous by unique id, contiguous shards, the ring exchange driven by
the step-loop launch's progress counter (graph and eager), the per-step-launch form, single eager steps.  What is
checked: every rank ends up with BOTH shards' records of the last step in global environment order, equal to what each
shard computes on its own.  (The reference has no multi-device path: SURVEY 8e.)
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
rank, world, mode, name, n_local, steps, idhex, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7], sys.argv[8]
sys.path.insert(0, os.environ["TDS_ROOT"])
import torch
import tds_amd
from tds_amd import hip_backend
m = tds_amd.load_model(name)
assert hip_backend.HipShard.rccl_version() == 99999, "the stub must be the loaded librccl"
g = np.load(os.path.join(os.environ["TDS_ROOT"], "tests", "golden", name + ".npz"))
rng = np.random.default_rng(1234)
idx = rng.integers(0, g["x"].shape[0], world * n_local)
xg = g["x"][idx]                                   # the GLOBAL batch, the same on every rank
ag = rng.uniform(-0.4, 0.4, (4, world * n_local, m.action_dim))
lo, hi = rank * n_local, (rank + 1) * n_local
sh = hip_backend.HipShard(m, world * n_local, rank=rank, world=world, device=0, dtype="f64",
                          unique_id=bytes.fromhex(idhex), wire_dtype="f32", options={"shard_chunk": 64})
sim = sh.sim
sim.x.copy_(torch.from_numpy(xg[lo:hi]).cuda())
acts = torch.from_numpy(ag[:, lo:hi]).cuda().contiguous()
if mode == "single":
    for k in range(steps):
        sh.step(acts[k % 4], 1)
else:
    done = 0
    for c in (steps - steps // 3, steps // 3):     # two calls: graphs for two shapes, ring halves carried over
        if c:
            sh.step_many(acts, c, first_block=done % 4)
            done += c
gat = sh.gathered().clone()
form = sh.exchange_form()
# ring exchange: EVERY slot of the most recently submitted launch, not only the last one
backs = []
if mode != "single" and form in ("peer_stores", "rccl_group_after_launch", "rccl_per_slot"):
    last = steps // 3 if steps // 3 else steps
    backs = [sh.gathered_step(b).clone() for b in range(min(last, 64))]
sh.flush()
torch.cuda.synchronize()
# what this shard computes on its own: the same steps on a plain handle
ref = hip_backend.HipSim(m, n_local, dtype="f64")
ref.x.copy_(torch.from_numpy(xg[lo:hi]).cuda())
obs = torch.zeros((n_local, ref.obs_dim + 2), dtype=torch.float64, device="cuda")
# (bitwise comparison: the plain handle takes the form the shard took — single steps where the shard exchanges per step,
#  e.g. the 16-lane quadruped kernel, whose step-loop compilation rounds differently from its straight-line one)
if mode == "single" or form == "rccl_per_step" or not ref.step_many_is_loop(steps) or os.environ.get("TDS_HIP_SHARD_RING") == "0":
    for k in range(steps):
        ref.step(acts[k % 4], 1, obs)
else:
    ring = torch.zeros((steps, n_local, ref.obs_dim + 2), dtype=torch.float64, device="cuda")
    # (a progress counter selects the build option exchange_w2 names, the one exchange launches use: same build, same bits)
    ref.step_many_rings(acts, steps, ring, None, progress=torch.zeros(steps, dtype=torch.int64, device="cuda"))
    obs = ring[-1]
torch.cuda.synchronize()
extra = {}
if backs:
    extra["backs"] = torch.stack(backs).cpu().numpy()                         # [b][world][n_local][w] float
    extra["local_backs"] = torch.stack([ring[steps - 1 - b] for b in range(len(backs))]).to(torch.float32).cpu().numpy()
np.savez(out, gathered=gat.cpu().numpy(), local=obs.to(torch.float32).cpu().numpy(), x=sim.x.cpu().numpy(), xref=ref.x.cpu().numpy(),
         form=np.array(form), peers=np.array(sh.peer_count()), **extra)
sh.close()
'''


def _stub(tmp_path_factory):
    d = tmp_path_factory.mktemp("stub_rccl")
    so = os.path.join(str(d), "libstub_rccl.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-o", so, os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp"), "-L/opt/rocm/lib",
                           "-lamdhip64", "-lrt", "-lpthread"])
    return so


@pytest.fixture(scope="module")
def stub_lib(tmp_path_factory):
    return _stub(tmp_path_factory)


RCCL = {"TDS_HIP_SHARD_PEER": "0"}  # the forms of the exchange that go through ncclAllGather (here: the stub)


@pytest.mark.parametrize("mode,name,steps,env,want_form", [
    # PEER STORES (the default): each process maps the other's gathered ring and flag array through hipIpcOpenMemHandle — real
    # IPC between two processes, here onto the one GPU — and its step-loop launch stores every record on both "ranks"
    ("many", "ant", 75, {"TDS_HIP_SHARD_PEER": "2"}, "peer_stores"),
    ("many", "ant", 75, {}, "peer_stores"),
    ("many", "ant", 75, {"TDS_HIP_EXCHANGE_FIELDS": "1"}, "peer_stores"),   # only [reward | done] travel to the peer
    ("many", "ant", 75, {"_n_local": "201"}, "peer_stores"),                # a ragged last wavefront: lane-per-component stores instead of whole rows
    ("many", "ant", 75, {"_n_local": "201", "TDS_HIP_EXCHANGE_FIELDS": "1"}, "peer_stores"),
    ("many", "pendulum5", 30, {}, "peer_stores"),                           # a world without contacts (the serial-chain kernel)
    # the STAGED form of the same exchange (option shard_peer_copy): the launch writes its own ring only, the communication
    # stream copies the launch's slots into the peer's ring (one strided device-to-device copy) and raises the flags
    ("many", "ant", 75, {"TDS_HIP_SHARD_PEER": "2", "TDS_HIP_SHARD_PEER_COPY": "1"}, "peer_copy"),
    ("many", "ant", 75, {"_n_local": "201", "TDS_HIP_SHARD_PEER_COPY": "1"}, "peer_copy"),
    ("many", "pendulum5", 30, {"TDS_HIP_SHARD_PEER_COPY": "1"}, "peer_copy"),
    ("many", "ant", 75, dict(RCCL), "rccl_group_after_launch"),             # ring exchange, two-wavefront build, slots sent behind the launch as one group
    ("many", "ant", 75, dict(RCCL, TDS_HIP_EXCHANGE_W2="0"), "rccl_per_slot"),  # ... the one-wave build: per-slot counters, a slot sent while the launch runs
    ("many", "ant", 75, dict(RCCL, TDS_HIP_SHARD_GRAPH="1"), "rccl_group_after_launch"),  # ring exchange, one hipGraph per step-loop launch
    ("many", "ant", 75, dict(RCCL, TDS_HIP_RING_NOFENCE="0", TDS_HIP_EXCHANGE_W2="0"), "rccl_per_slot"),  # records made visible by a release fence instead of write-through stores
    ("many", "pendulum5", 30, dict(RCCL), "rccl_per_slot"),                 # a world without contacts through RCCL
    ("many", "ant", 12, {"TDS_HIP_SHARD_RING": "0"}, "rccl_per_step"),      # per-step launches + exchanges from one hipGraph
    ("many", "laikago", 9, {}, "rccl_per_step"),                            # a model whose step_many is not the step-loop form
    ("single", "ant", 7, {}, "rccl_per_step"),                              # tds_hip_shard_step, one call per step
])
def test_two_ranks_on_one_gpu(mode, name, steps, env, want_form, built, stub_lib, tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world, n_local = 2, int(env.get("_n_local", 200))
    env = {k: v for k, v in env.items() if not k.startswith("_")}
    ident = ("/tds_stub_%d_%s_%d" % (os.getpid(), name, steps)).encode().ljust(128, b"\0")
    e = dict(os.environ)
    e.update(env)
    e["TDS_HIP_RCCL_LIB"] = stub_lib
    e["TDS_ROOT"] = ROOT
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, str(r), str(world), mode, name, str(n_local), str(steps),
                               ident.hex(), outs[r]], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    r = [np.load(o) for o in outs]
    w = r[0]["local"].shape[1]
    want = np.concatenate([r[0]["local"], r[1]["local"]])  # global environment order = rank order
    rd_only = env.get("TDS_HIP_EXCHANGE_FIELDS") == "1"
    for k in range(world):
        assert str(r[k]["form"]) == want_form, (k, str(r[k]["form"]), logs[k])
        assert int(r[k]["peers"]) == (1 if want_form in ("peer_stores", "peer_copy") else -1)
        got = r[k]["gathered"].reshape(-1, w)
        assert got.shape == want.shape
        if rd_only:  # the OTHER rank's block carries [reward | done] only (its observation columns are never written)
            other = slice((1 - k) * n_local, (2 - k) * n_local)
            assert np.array_equal(got[other, -2:], want[other, -2:], equal_nan=True) and (got[other, :-2] == 0).all()
            got[other, :-2] = want[other, :-2]
        if "backs" in r[k].files:  # every slot of the last launch, both ranks' blocks
            lb = np.concatenate([r[0]["local_backs"], r[1]["local_backs"]], axis=1)  # [b][world n_local][w]
            gb = r[k]["backs"].reshape(lb.shape).copy()
            if rd_only:
                gb[:, other, :-2] = lb[:, other, :-2]
            assert np.array_equal(gb, lb, equal_nan=True), k
        # the records crossed "the wire" as floats; ring form and its reference come from the same step-loop build, the
        # per-step forms from the same straight-line build: equal bit for bit
        # (random states driven by random actions: a few environments leave the finite range on the way)
        assert np.array_equal(got, want, equal_nan=True), (k, np.nanmax(np.abs(got - want)))
        assert np.isfinite(want).mean() > 0.9
        assert np.array_equal(r[k]["x"], r[k]["xref"], equal_nan=True)
