"""FIRST CONTACT with more than one GPU: the default N > 1 path of the C shard layer (csrc/tds_shard.hip) with one process per
GPU, device = rank, the REAL librccl for the handle exchange and the peer-store exchange over xGMI (IPC-mapped gathered rings,
system-scope record stores, arrival counters, flags raised on every rank).

A 1-GPU box SKIPS these tests — loudly: nothing here can run there (tests/test_shard_two_ranks_one_gpu.py covers the protocol with
two processes on one GPU through the same IPC calls; tests/test_multi_gpu.py pins the gathered slots on the reference on one
rank).  On a node with 2 / 4 / 8 GPUs every variant runs with as many ranks as there are devices.  What is checked:

  * every gathered slot of the last launch, on EVERY rank, both halves of the ring in use: bit for bit what each shard computes
    alone on a plain handle (same launch form, same build), in global environment order;
  * rank 0's own block of every slot against the REAL reference (oracle/_ref/libtds_ref.so) started from the state the slot
    before holds (per-step resync, 1e-6 relative per step: north_star);
  * a SOAK: >= 10 000 steps in calls of uneven length over ragged shards (a last wavefront that is not full), both ring halves
    reused >= 20 times — a stale line (a slot read from this GPU's cache instead of what the peer stored) or a torn record (a flag
    that overtook its records) shows up as a slot that differs from the shard's own recomputation;
  * the ordering A/B of the protocol in one run: option shard_peer_release = 1 (system-scope release fences in front of the
    arrival counts and the flag stores) must give the same bits as the default (vmcnt(0) + relaxed stores);
  * the fallbacks: option shard_peer = 0 (ncclAllGather of the launch's slots) and exchange_fields = 1 ([reward | done] only).

The reference has no multi-device path; contract: SURVEY.md 8(e), include/tds_hip.h (tds_hip_shard_*).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
rank, world, name, n_local, calls, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), json.loads(sys.argv[5]), sys.argv[6], sys.argv[7]
sys.path.insert(0, os.environ["TDS_ROOT"])
import time
import torch
import tds_amd
from tds_amd import hip_backend
torch.cuda.set_device(rank)
m = tds_amd.load_model(name)
# the communicator's id: rank 0 makes it (the REAL librccl), the others pick it up from a file
if rank == 0:
    uid = hip_backend.HipShard.unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120, "rank 0 never published the communicator id"
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
assert hip_backend.HipShard.rccl_version() != 99999, "the REAL librccl, not the tests' stub"
rng = np.random.default_rng(4321)
n_glob = world * n_local
ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
xg = np.zeros((n_glob, m.input_dim))
if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION:
    xg[:, 2] = 0.48
    xg[:, 6:m.dof_q] = ip + 0.05 * rng.uniform(-1, 1, (n_glob, m.dof_q - 6))
    xg[:, -3:] = [15, 0.3, 3] if name.startswith("ant") else [100, 2, 50]
else:
    xg[:, :m.dof_q] = rng.uniform(-1, 1, (n_glob, m.dof_q))
ag = rng.uniform(-0.4, 0.4, (8, n_glob, m.action_dim))
lo, hi = rank * n_local, (rank + 1) * n_local
opts = {"shard_chunk": int(os.environ.get("_CHUNK", "64"))}
sh = hip_backend.HipShard(m, n_glob, rank=rank, world=world, device=rank, dtype="f64", unique_id=uid, wire_dtype="f32", options=opts)
sim = sh.sim
sim.x.copy_(torch.from_numpy(xg[lo:hi]).cuda())
acts = torch.from_numpy(ag[:, lo:hi]).cuda().contiguous()
ref = hip_backend.HipSim(m, n_local, device=rank, dtype="f64")
ref.x.copy_(torch.from_numpy(xg[lo:hi]).cuda())
w = ref.obs_dim + 2
done, bad, checked = 0, [], 0
last_backs = last_ring = None
x_before_last = None
for c in calls:
    x_before_last = ref.x.clone()
    sh.step_many(acts, c, first_block=done % 8)
    # (tds_hip_shard_gathered_step reaches back inside the most recently submitted LAUNCH: a call is cut into launches of
    #  shard_chunk steps and a remainder — csrc/tds_shard_plan.h)
    in_last_launch = (c - 1) % opts["shard_chunk"] + 1
    backs = [sh.gathered_step(b).clone() for b in range(in_last_launch)]      # [world][n_local][w] float, newest first
    ring = torch.zeros((c, n_local, w), dtype=torch.float64, device="cuda")
    ref.step_many_rings(acts, c, ring, None, first_block=done % 8, progress=torch.zeros(c, dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    # this rank's block of every slot == its own recomputation; the peers' blocks are checked by the parent (it has all ranks' files
    # for the LAST call) and here against all-gathered copies for EVERY call
    for b, gb in enumerate(backs):
        mine = ring[c - 1 - b].to(torch.float32)
        checked += 1
        if not torch.equal(gb[rank].view(torch.int32), mine.view(torch.int32)):
            bad.append((done, b, "own block"))
    last_backs, last_ring = backs, ring
    done += c
form = sh.exchange_form()
peers = sh.peer_count()
sh.flush()
torch.cuda.synchronize()
np.savez(out, backs=torch.stack(last_backs).cpu().numpy(), local_backs=torch.stack([last_ring[len(last_ring) - 1 - b] for b in range(len(last_backs))]).to(torch.float32).cpu().numpy(),
         x=sim.x.cpu().numpy(), xref=ref.x.cpu().numpy(), form=np.array(form), peers=np.array(peers), bad=np.array(len(bad)), checked=np.array(checked),
         bad_list=np.array(str(bad[:8])), y_last=ref.y.cpu().numpy(), x_before_last=x_before_last.cpu().numpy(), ring_last=last_ring.cpu().numpy(),
         acts=ag[:, lo:hi], last_first=np.array((done - calls[-1]) % 8))
sh.close()
'''


def _device_count():
    sys.path.insert(0, ROOT)
    from tds_amd import hip_backend

    return int(hip_backend.lib().tds_hip_device_count())


def _run(world, name, n_local, calls, env, tmp_path, timeout=1500):
    import json

    e = dict(os.environ)
    e.update(env)
    e["TDS_ROOT"] = ROOT
    e["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    e.pop("TDS_HIP_RCCL_LIB", None)
    idfile = str(tmp_path / "nccl_id")
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, str(r), str(world), name, str(n_local), json.dumps(calls), idfile, outs[r]],
                              env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace")[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    return [np.load(o) for o in outs], logs


def _check_all_blocks(r, world, n_local, rd_only=False):
    """every rank's copy of every slot of the last call == the ranks' own recomputations, in global environment order"""
    lb = np.concatenate([r[k]["local_backs"] for k in range(world)], axis=1)  # [b][world n_local][w]
    for k in range(world):
        gb = r[k]["backs"].reshape(lb.shape).copy()
        if rd_only:  # the OTHER ranks' blocks carry [reward | done] only
            for j in range(world):
                if j != k:
                    blk = slice(j * n_local, (j + 1) * n_local)
                    assert (gb[:, blk, :-2] == 0).all(), (k, j)
                    gb[:, blk, :-2] = lb[:, blk, :-2]
        assert np.array_equal(gb.view(np.int32), lb.view(np.int32)), (k, "a gathered slot differs from the shards' own records")
        assert int(r[k]["bad"]) == 0, (k, str(r[k]["bad_list"]))
        assert np.array_equal(r[k]["x"], r[k]["xref"], equal_nan=True)


WORLDS = [2, 4, 8]


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("variant,env,want_form", [
    ("peer_stores", {"TDS_HIP_SHARD_PEER": "2"}, "peer_stores"),                     # the default, required (an error instead of the fallback)
    ("peer_stores_release_fences", {"TDS_HIP_SHARD_PEER": "2", "TDS_HIP_SHARD_PEER_RELEASE": "1"}, "peer_stores"),
    ("reward_done_only", {"TDS_HIP_SHARD_PEER": "2", "TDS_HIP_EXCHANGE_FIELDS": "1"}, "peer_stores"),
    ("staged_copies", {"TDS_HIP_SHARD_PEER": "2", "TDS_HIP_SHARD_PEER_COPY": "1"}, "peer_copy"),  # copy engines instead of stores from the kernel
    ("rccl", {"TDS_HIP_SHARD_PEER": "0"}, None),                                       # ncclAllGather of the launch's slots
])
def test_every_gathered_slot_on_every_rank_over_the_fabric(world, variant, env, want_form, built, tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    nd = _device_count()
    if nd < world:
        pytest.skip(f"MULTI-GPU PATH NOT EXERCISED: {world} ranks need {world} GPUs, this box has {nd} "
                    f"(the N > 1 exchange over xGMI has then never run here)")
    n_local = 4096
    r, logs = _run(world, "ant", n_local, [70, 45], env, tmp_path)
    for k in range(world):
        if want_form:
            assert str(r[k]["form"]) == want_form, (k, str(r[k]["form"]), logs[k])
            assert int(r[k]["peers"]) == world - 1
    _check_all_blocks(r, world, n_local, rd_only=variant == "reward_done_only")
    # rank 0's block of every slot of the last call against the REAL reference, each step from the state the slot before holds
    from test_hip_parity import _reference_stepper
    from conftest import rel_err
    import tds_amd

    m = tds_amd.load_model("ant")
    ref_step, what = _reference_stepper("ant", n_local)
    ring = r[0]["ring_last"]  # [c][n_local][w] double: obs | reward | done of every step of the last call
    nq, nd_, adim = m.dof_q, m.dof_qd, m.action_dim
    x = r[0]["x_before_last"].copy()
    worst = 0.0
    for k in range(ring.shape[0]):
        x[:, nq + nd_:nq + nd_ + adim] = r[0]["acts"][(int(r[0]["last_first"]) + k) % 8]
        y = ref_step(x)
        got = ring[k][:, 2:nq + nd_]
        worst = max(worst, rel_err(got, y[:, 2:nq + nd_]))
        x[:, :nq + nd_] = np.concatenate([y[:, :2], got], axis=1)  # resync on the device's own state (x, y: the reference's)
    print(f"{world} ranks [{variant}]: rank 0's slots vs {what}: worst per-step rel err {worst:.3e}")
    assert worst < 1e-6


@pytest.mark.parametrize("world", WORLDS)
def test_soak_ragged_shards_ring_reuse(world, built, tmp_path):
    """>= 10 000 steps, uneven calls, 4093 environments per rank (a ragged last wavefront: lane-per-component peer stores beside
    whole-row ones), a 32-step ring half: every half reused ~ 160 times"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    nd = _device_count()
    if nd < world:
        pytest.skip(f"MULTI-GPU PATH NOT EXERCISED: {world} ranks need {world} GPUs, this box has {nd}")
    calls = []
    rng = np.random.default_rng(5)
    while sum(calls) < 10000:
        calls.append(int(rng.integers(1, 200)))
    r, logs = _run(world, "ant", 4093, calls, {"TDS_HIP_SHARD_PEER": "2", "_CHUNK": "32"}, tmp_path, timeout=3000)
    for k in range(world):
        assert str(r[k]["form"]) == "peer_stores", (k, str(r[k]["form"]))
        assert int(r[k]["checked"]) >= 2000
    _check_all_blocks(r, world, 4093)


def test_the_worker_of_these_tests_on_one_rank(built, tmp_path):
    """(runs on every box) the same worker with ONE rank on device 0 — the real librccl's one-rank communicator, the peer-store
    launch with no peer: what the fabric tests execute per rank is exercised wherever the suite runs"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r, logs = _run(1, "ant", 2045, [70, 45, 9], {"_CHUNK": "32"}, tmp_path)
    assert str(r[0]["form"]) == "peer_stores", (str(r[0]["form"]), logs[0])
    _check_all_blocks(r, 1, 2045)
    assert int(r[0]["checked"]) == 6 + 13 + 9  # (the steps of each call's last launch: 70 = 32 + 32 + 6, 45 = 32 + 13, 9)


def test_the_multi_gpu_tests_say_when_they_did_not_run(built):
    """(always runs) one line in the log about what this box could exercise"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    nd = _device_count()
    print(f"devices on this box: {nd}; fabric tests run with world in {[w for w in WORLDS if w <= nd] or 'NONE (1 GPU)'}")
    assert nd >= 1
