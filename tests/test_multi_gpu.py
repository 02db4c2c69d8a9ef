"""GPU: the multi-GPU shard layer of the C ABI (tds_hip_shard_*: contiguous environment shards + one RCCL all-gather
of the [obs | reward | done] records per policy step) and the hipGraph step loop (tds_hip_step_many).

A 1-GPU box exercises the whole path with ONE rank (RCCL communicator of size 1, communication stream, ring of record
blocks, wire conversion); the 2-rank test runs wherever tds_hip_device_count() >= 2 and skips loudly otherwise.  The
the N > 1 host protocol and the shard arithmetic on CPU are covered by tests/test_rank_protocol_gloo.py (gloo, world_size 2) and tests/test_shard_plan.py."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

import tds_amd
from conftest import rel_err
from tds_amd import hip_backend

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def _start(m, n, seed=5):
    g = np.load(os.path.join(GOLDEN, "ant.npz"))
    rng = np.random.default_rng(seed)
    x = g["x"][rng.integers(0, g["x"].shape[0], n)].copy()
    acts = rng.uniform(-0.4, 0.4, (6, n, m.action_dim))
    return x, acts


@pytest.mark.parametrize("with_rccl", [False, True])
@pytest.mark.parametrize("wire", ["f32", "f64"])
def test_one_rank_shard_equals_plain_stepping(with_rccl, wire, built):
    torch = _torch()
    if with_rccl and hip_backend.HipShard.rccl_version() == 0:
        pytest.skip("librccl cannot be loaded on this machine")
    m = tds_amd.load_model("ant")
    n = 256
    x, acts = _start(m, n)
    ref = hip_backend.HipSim(m, n)
    ref.x.copy_(torch.from_numpy(x).cuda())
    uid = hip_backend.HipShard.unique_id() if with_rccl else None
    sh = hip_backend.HipShard(m, n, rank=0, world=1, device=0, dtype="f64", unique_id=uid, wire_dtype=wire)
    sh.sim.x.copy_(torch.from_numpy(x).cuda())
    obs = torch.zeros((n, ref.obs_dim + 2), dtype=torch.float64, device="cuda")
    for k in range(6):
        a = torch.from_numpy(acts[k]).cuda()
        ref.step(a, 1, obs)
        sh.step(a)
        got = sh.gathered()  # (the current stream waits for the exchange)
        assert tuple(got.shape) == (1, 1, n, ref.obs_dim + 2)
        want = obs if wire == "f64" else obs.float()
        assert got.dtype == want.dtype
        assert torch.equal(got[0, 0], want), k
    sh.flush()
    assert torch.equal(sh.sim.x, ref.x) and torch.equal(sh.sim.y, ref.y)
    sh.close()


def test_blocked_exchange_carries_every_step(built):
    """steps_per_exchange = 4 (the pipelined form): one all-gather carries the records of four consecutive steps"""
    torch = _torch()
    m = tds_amd.load_model("ant")
    n = 128
    x, acts = _start(m, n, seed=8)
    ref = hip_backend.HipSim(m, n)
    ref.x.copy_(torch.from_numpy(x).cuda())
    uid = hip_backend.HipShard.unique_id() if hip_backend.HipShard.rccl_version() else None
    sh = hip_backend.HipShard(m, n, unique_id=uid, wire_dtype="f64", block=4)
    sh.sim.x.copy_(torch.from_numpy(x).cuda())
    recs = []
    for k in range(6):
        a = torch.from_numpy(acts[k]).cuda()
        o = torch.zeros((n, ref.obs_dim + 2), dtype=torch.float64, device="cuda")
        ref.step(a, 1, o)
        recs.append(o)
        sh.step(a)
        if k == 3:
            got = sh.gathered()
            assert tuple(got.shape) == (1, 4, n, ref.obs_dim + 2)
            for j in range(4):
                assert torch.equal(got[0, j], recs[j]), j
    sh.flush()  # the partial block (steps 4, 5) travels at the flush
    got = sh.gathered()
    assert torch.equal(got[0, 0], recs[4]) and torch.equal(got[0, 1], recs[5])
    sh.close()


def test_float_records_travel_unconverted(built):
    torch = _torch()
    m = tds_amd.load_model("ant")
    n = 64
    x, acts = _start(m, n, seed=9)
    sh = hip_backend.HipShard(m, n, dtype="mixed", wire_dtype="f64")  # (never widened on the wire)
    assert sh.wire_torch_dtype == torch.float32
    sh.sim.x.copy_(torch.from_numpy(x).float().cuda())
    ref = hip_backend.HipSim(m, n, dtype="mixed")
    ref.x.copy_(torch.from_numpy(x).float().cuda())
    obs = torch.zeros((n, ref.obs_dim + 2), dtype=torch.float32, device="cuda")
    a = torch.from_numpy(acts[0]).float().cuda()
    ref.step(a, 1, obs)
    sh.step(a)
    assert torch.equal(sh.gathered()[0, 0], obs)
    sh.close()


def test_two_ranks_one_process_rccl(built):
    """Two GPUs driven by one process (tds_hip_shard_create_all + tds_hip_shard_group_step): both ranks end up with
    the records of the whole batch in global environment order."""
    torch = _torch()
    L = hip_backend.lib()
    if L.tds_hip_device_count() < 2:
        pytest.skip("SKIPPED LOUDLY: this box has %d GPU(s); the 2-rank RCCL exchange needs 2" % L.tds_hip_device_count())
    m = tds_amd.load_model("ant")
    n_global = 512
    x, acts = _start(m, n_global, seed=21)
    devs = (C.c_int * 2)(0, 1)
    hs = (C.c_void_p * 2)()
    rc = L.tds_hip_shard_create_all(C.byref(m), n_global, 2, devs, tds_amd.TDS_DTYPE_F64, tds_amd.TDS_DTYPE_F64, hs)
    assert rc == 0, L.tds_hip_last_error()
    half = n_global // 2
    ref = hip_backend.HipSim(m, n_global)
    ref.x.copy_(torch.from_numpy(x).cuda())
    sims = []
    for r in range(2):
        assert L.tds_hip_shard_first_env(hs[r]) == r * half
        s = hip_backend.HipSim(m, half, device=r, _handle=C.c_void_p(L.tds_hip_shard_sim(hs[r])), _owner=object())
        with torch.cuda.device(r):
            s.x.copy_(torch.from_numpy(x[r * half:(r + 1) * half]).to(f"cuda:{r}"))
        sims.append(s)
    obs = torch.zeros((n_global, ref.obs_dim + 2), dtype=torch.float64, device="cuda:0")
    for k in range(4):
        ref.step(torch.from_numpy(acts[k]).cuda(), 1, obs)
        a_dev = [torch.from_numpy(acts[k][r * half:(r + 1) * half]).to(f"cuda:{r}") for r in range(2)]
        ap = (C.c_void_p * 2)(a_dev[0].data_ptr(), a_dev[1].data_ptr())
        assert L.tds_hip_shard_group_step(hs, 2, ap, 1) == 0, L.tds_hip_last_error()
        for r in range(2):
            assert L.tds_hip_shard_flush(hs[r]) == 0
            ptr, blk = C.c_void_p(), C.c_int()
            assert L.tds_hip_shard_gathered(hs[r], None, C.byref(ptr), C.byref(blk)) == 0
            with torch.cuda.device(r):
                host = hip_backend.wrap_device_pointer(ptr.value, (n_global, ref.obs_dim + 2), torch.float64, r).cpu().numpy()
            assert np.array_equal(host, obs.cpu().numpy()), (k, r)
    for r in range(2):
        sims[r].h = None
        L.tds_hip_shard_destroy(hs[r])


def test_create_restores_the_callers_device(built):
    torch = _torch()
    m = tds_amd.load_model("ant")
    before = torch.cuda.current_device()
    sim = hip_backend.HipSim(m, 16, device=0)
    assert torch.cuda.current_device() == before
    assert hip_backend.lib().tds_hip_device(sim.h) == 0 and hip_backend.lib().tds_hip_record_bytes(sim.h) == 8


@pytest.mark.parametrize("dtype", ["f64", "mixed"])
def test_step_many_graph_equals_eager_launches(dtype, built, monkeypatch):
    torch = _torch()
    monkeypatch.setenv("TDS_HIP_STEP_MANY_LOOP", "0")  # (the graph form; the one-launch form: the tests below)
    m = tds_amd.load_model("ant")
    n = 192
    x, acts = _start(m, n, seed=13)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    a = torch.from_numpy(acts).to(tdt).cuda().contiguous()  # [6, n, adim]
    eager, graph = hip_backend.HipSim(m, n, dtype=dtype), hip_backend.HipSim(m, n, dtype=dtype)
    for s in (eager, graph):
        s.x.copy_(torch.from_numpy(x).to(tdt).cuda())
    oe = torch.zeros((n, eager.obs_dim + 2), dtype=tdt, device="cuda")
    og = torch.zeros_like(oe)
    K = 17
    for k in range(K):
        eager.step(a[(2 + k) % 6], 1, oe)
    graph.step_many_prepare(a, K, og, first_block=2)  # nothing runs
    assert torch.equal(graph.x, torch.from_numpy(x).to(tdt).cuda())
    graph.step_many(a, K, og, first_block=2)
    torch.cuda.synchronize()
    assert torch.equal(graph.x, eager.x) and torch.equal(graph.y, eager.y) and torch.equal(og, oe)
    # replay of the cached graph continues from the new state
    for k in range(K):
        eager.step(a[(2 + k) % 6], 1, oe)
    graph.step_many(a, K, og, first_block=2)
    torch.cuda.synchronize()
    assert torch.equal(graph.x, eager.x) and torch.equal(og, oe)
    # timing covers the whole call
    graph.set_timing(True)
    graph.step_many(a, K, og, first_block=2)
    assert graph.last_kernel_ms() > 0.0


@pytest.mark.parametrize("chains,n", [("1", 100), ("3", 100), ("8", 100), ("8", 20), ("2", 2052)])
def test_step_many_environment_chains_equal_whole_batch_launches(chains, n, built, monkeypatch):
    """The graph's C environment chains (contiguous ranges, one branch each) leave bit for bit the records of K
    whole-batch launches, whatever C and however unevenly the workgroups divide (2052 envs: slab for surplus rows
    in play, one-wavefront / two-wavefront form chosen on the whole batch)."""
    torch = _torch()
    monkeypatch.setenv("TDS_HIP_STEP_MANY_LOOP", "0")
    monkeypatch.setenv("TDS_HIP_GRAPH_CHAINS", chains)
    m = tds_amd.load_model("ant")
    x, acts = _start(m, n, seed=5)
    a = torch.from_numpy(acts).cuda().contiguous()
    eager, graph = hip_backend.HipSim(m, n), hip_backend.HipSim(m, n)
    for s in (eager, graph):
        s.x.copy_(torch.from_numpy(x).cuda())
    oe = torch.zeros((n, eager.obs_dim + 2), dtype=torch.float64, device="cuda")
    og = torch.zeros_like(oe)
    K = 40
    for k in range(K):
        eager.step(a[k % 6], 1, oe)
    graph.step_many(a, K, og, first_block=0)
    torch.cuda.synchronize()

    def bits(t):  # (random states driven by random actions: a few environments diverge to NaN within 40 steps)
        return t.view(torch.int64)

    assert torch.equal(bits(graph.x), bits(eager.x)) and torch.equal(bits(graph.y), bits(eager.y))
    assert torch.equal(bits(og), bits(oe))


@pytest.mark.parametrize("name,dtype", [("pendulum5", "f64"), ("pendulum5", "mixed"), ("cartpole", "f64")])
def test_step_many_of_a_contact_free_world_is_one_loop_launch(name, dtype, built, monkeypatch):
    """Worlds without contact points run their K steps as ONE launch of the step-loop build, every step with its own
    action block (TdsStepCtl::act_pool): same records as K straight-line launches (to round-off: another build), and
    the forced graph form agrees too."""
    torch = _torch()
    m = tds_amd.load_model(name)
    n = 300
    rng = np.random.default_rng(21)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    x = np.zeros((n, m.input_dim))
    x[:, : m.dof_q] = rng.uniform(-0.5, 0.5, (n, m.dof_q))
    x[:, m.dof_q: m.dof_q + m.dof_qd] = rng.uniform(-0.5, 0.5, (n, m.dof_qd))
    acts = rng.uniform(-0.3, 0.3, (6, n, m.action_dim))
    a = torch.from_numpy(acts).to(tdt).cuda().contiguous()
    K = 37
    res = {}
    for form in ("eager", "loop", "graph"):
        if form == "graph":
            monkeypatch.setenv("TDS_HIP_STEP_MANY_LOOP", "0")
        sim = hip_backend.HipSim(m, n, dtype=dtype)
        sim.x.copy_(torch.from_numpy(x).to(tdt).cuda())
        obs = torch.zeros((n, sim.obs_dim + 2), dtype=tdt, device="cuda")
        if form == "eager":
            for k in range(K):
                sim.step(a[(4 + k) % 6], 1, obs)
        else:
            sim.step_many(a, K, obs, first_block=4)
        torch.cuda.synchronize()
        res[form] = (sim.x.double().cpu().numpy(), sim.y.double().cpu().numpy(), obs.double().cpu().numpy())
    # (float records: K launches round the state to float after every step, the loop keeps it in double in LDS and
    #  rounds once at the end — per step both are within the contract, the trajectories drift apart at float rounding)
    tol = 1e-9 if dtype == "f64" else 1e-3
    for form in ("loop", "graph"):
        for u, v in zip(res[form], res["eager"]):
            assert rel_err(u, v) < tol, form
    assert np.array_equal(res["graph"][0], res["eager"][0])  # (same build, same rounding points: bit for bit)


@pytest.mark.parametrize("dtype,n", [("f64", 1024), ("mixed", 1024), ("f64", 9000)])
def test_step_many_of_the_ant_is_one_loop_launch(dtype, n, built, monkeypatch):
    """Narrow kernels with contacts, up to three rounds of workgroups: the K steps run as ONE launch of the step-loop build
    (action block per step); against K straight-line launches (another build: to round-off) and against the graphs."""
    torch = _torch()
    m = tds_amd.load_model("ant")
    x, acts = _start(m, n, seed=17)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    a = torch.from_numpy(acts).to(tdt).cuda().contiguous()
    K = 25
    res = {}
    for form in ("eager", "loop", "graph"):
        if form == "graph":
            monkeypatch.setenv("TDS_HIP_STEP_MANY_LOOP", "0")
        sim = hip_backend.HipSim(m, n, dtype=dtype)
        assert sim.step_many_is_loop(K) == (form != "graph")
        sim.x.copy_(torch.from_numpy(x).to(tdt).cuda())
        obs = torch.zeros((n, sim.obs_dim + 2), dtype=tdt, device="cuda")
        if form == "eager":
            for k in range(K):
                sim.step(a[(2 + k) % 6], 1, obs)
        else:
            sim.step_many(a, K, obs, first_block=2)
        torch.cuda.synchronize()
        res[form] = (sim.x.double().cpu().numpy(), sim.y.double().cpu().numpy(), obs.double().cpu().numpy())
    ok = np.isfinite(res["eager"][0]).all(axis=1) & np.isfinite(res["loop"][0]).all(axis=1)
    assert ok.sum() > 0.9 * n
    # (float records: rounded once at the end instead of after every step — the trajectories drift apart at float
    #  rounding, amplified by 25 steps of contact dynamics)
    tol = 1e-7 if dtype == "f64" else 5e-2
    for u, v in zip(res["loop"], res["eager"]):
        assert rel_err(u[ok], v[ok]) < tol
    assert np.array_equal(res["graph"][0].view(np.int64), res["eager"][0].view(np.int64))


def test_step_many_tune_picks_a_chain_count_and_keeps_the_records(built, monkeypatch):
    torch = _torch()
    monkeypatch.setenv("TDS_HIP_STEP_MANY_LOOP", "0")
    m = tds_amd.load_model("ant")
    n = 512
    x, acts = _start(m, n, seed=9)
    a = torch.from_numpy(acts).cuda().contiguous()
    eager, graph = hip_backend.HipSim(m, n), hip_backend.HipSim(m, n)
    for s in (eager, graph):
        s.x.copy_(torch.from_numpy(x).cuda())
    c = graph.tune_step_many(a, probe_steps=12)  # 6 x 12 steps, every probe from action block 0
    assert 1 <= c <= 3
    for rep in range(6):
        for k in range(12):
            eager.step(a[k % 6])
    def bits(t):  # (random states driven by random actions: a few environments diverge to NaN on the way)
        return t.view(torch.int64)

    assert torch.equal(bits(graph.x), bits(eager.x))
    graph.step_many(a, 30, first_block=1)
    for k in range(30):
        eager.step(a[(1 + k) % 6])
    torch.cuda.synchronize()
    assert torch.equal(bits(graph.x), bits(eager.x)) and torch.equal(bits(graph.y), bits(eager.y))
    with pytest.raises(Exception):
        graph.set_graph_chains(9)


@pytest.mark.parametrize("with_rccl", [False, True])
@pytest.mark.parametrize("submit", ["graph", "eager"])
@pytest.mark.parametrize("wire", ["f32", "f64"])
def test_shard_ring_exchange_on_one_rank(with_rccl, submit, wire, built, monkeypatch):
    """tds_hip_shard_step_many where the K steps are step-loop launches (the Ant): every step's [obs | reward | done]
    record goes into a ring slot in the wire dtype and is all-gathered as soon as the running launch has counted its
    workgroups in for that step.  Against the same launches without the exchange (tds_hip_step_many_rings): the gathered
    records of the last step, the state and the y record, bit for bit; launches of 64 + 11 steps, then a replay."""
    torch = _torch()
    if with_rccl and hip_backend.HipShard.rccl_version() == 0:
        pytest.skip("librccl cannot be loaded on this machine")
    if submit == "graph":
        monkeypatch.setenv("TDS_HIP_SHARD_GRAPH", "1")
    m = tds_amd.load_model("ant")
    n = 1000
    x, acts = _start(m, n, seed=41)
    a = torch.from_numpy(acts).cuda().contiguous()
    uid = hip_backend.HipShard.unique_id() if with_rccl else None
    # (default: 256 steps per launch; shard_peer = 0: the RCCL forms of the exchange — the default, peer stores, has its own tests below)
    sh = hip_backend.HipShard(m, n, unique_id=uid, wire_dtype=wire, options={"shard_chunk": 64, "shard_peer": 0})
    ref = hip_backend.HipSim(m, n)
    for s in (sh.sim, ref):
        s.x.copy_(torch.from_numpy(x).cuda())
    K = 75
    wdt = torch.float32 if wire == "f32" else torch.float64
    ring = torch.zeros((K, n, ref.obs_dim + 2), dtype=wdt, device="cuda")
    assert sh.sim.step_many_is_loop(K)
    for rep in range(3):
        if rep == 0:
            sh.step_many(a, K, first_block=1, prepare_only=True)
            assert torch.equal(sh.sim.x, torch.from_numpy(x).cuda())
        sh.step_many(a, K, first_block=1)
        got = sh.gathered().clone()
        # (a launch that is handed a progress counter takes the build option exchange_w2 names — since round 4 the
        #  two-wavefront build, the one N = 1 runs — on both sides: bit for bit)
        ref.step_many_rings(a, K, ring, None, first_block=1, progress=torch.zeros(K, dtype=torch.int64, device="cuda"))
        sh.flush()
        torch.cuda.synchronize()
        assert tuple(got.shape) == (1, 1, n, ref.obs_dim + 2) and got.dtype == wdt
        assert torch.equal(got[0, 0].view(torch.int32 if wire == "f32" else torch.int64),
                           ring[-1].view(torch.int32 if wire == "f32" else torch.int64)), rep
        assert torch.equal(sh.sim.x.view(torch.int64), ref.x.view(torch.int64)), rep
        assert torch.equal(sh.sim.y.view(torch.int64), ref.y.view(torch.int64)), rep
    # single eager steps afterwards use the per-step exchange and continue from the same state
    o = torch.zeros((n, ref.obs_dim + 2), dtype=torch.float64, device="cuda")
    sh.step(a[0])
    ref.step(a[0], 1, o)
    assert torch.equal(sh.gathered()[0, 0].view(torch.int32 if wire == "f32" else torch.int64),
                       o.to(wdt).view(torch.int32 if wire == "f32" else torch.int64))
    sh.close()


@pytest.mark.parametrize("with_rccl", [False, True])
def test_shard_step_many_graph_equals_eager_exchange(with_rccl, built, monkeypatch):
    """the two-stream step + all-gather pattern captured into one hipGraph (RCCL all-gathers as graph nodes): the
    per-step-launch form (models whose K steps are not one step-loop launch; forced here for the Ant)"""
    torch = _torch()
    monkeypatch.setenv("TDS_HIP_SHARD_RING", "0")
    if with_rccl and hip_backend.HipShard.rccl_version() == 0:
        pytest.skip("librccl cannot be loaded on this machine")
    m = tds_amd.load_model("ant")
    n = 256
    x, acts = _start(m, n, seed=31)
    a = torch.from_numpy(acts).cuda().contiguous()
    uid = hip_backend.HipShard.unique_id() if with_rccl else None
    eager = hip_backend.HipShard(m, n, unique_id=uid, wire_dtype="f64")
    uid2 = hip_backend.HipShard.unique_id() if with_rccl else None
    graph = hip_backend.HipShard(m, n, unique_id=uid2, wire_dtype="f64")
    for s in (eager, graph):
        s.sim.x.copy_(torch.from_numpy(x).cuda())
    K = 11
    for rep in range(3):  # first call captures, later calls replay the cached graph
        for k in range(K):
            eager.step(a[(1 + k) % 6])
        if rep == 0:
            graph.step_many(a, K, first_block=1, prepare_only=True)
            assert torch.equal(graph.sim.x, torch.from_numpy(x).cuda())
        graph.step_many(a, K, first_block=1)
        ge, gg = eager.gathered().clone(), graph.gathered().clone()
        torch.cuda.synchronize()
        assert torch.equal(graph.sim.x, eager.sim.x) and torch.equal(gg, ge), rep
    # eager steps after a graph launch continue the same ring
    eager.step(a[0])
    graph.step(a[0])
    assert torch.equal(graph.gathered(), eager.gathered())
    eager.close()
    graph.close()


def test_shard_step_many_with_auto_reset_steps_eagerly(built):
    """with auto-reset on, the K steps of tds_hip_shard_step_many are K auto-reset steps + exchanges (the reset pool's
    refill passes are host-driven, so nothing is captured): same records as K tds_hip_shard_step calls"""
    torch = _torch()
    m = tds_amd.load_model("ant")
    n = 192
    x, acts = _start(m, n, seed=37)
    x[::3, 2] = 0.27  # torso at the done threshold: resets within a few steps
    a = torch.from_numpy(acts).cuda().contiguous()
    one, many = hip_backend.HipShard(m, n, wire_dtype="f64"), hip_backend.HipShard(m, n, wire_dtype="f64")
    for s in (one, many):
        s.sim.set_auto_reset(True, 11)
        s.sim.x.copy_(torch.from_numpy(x).cuda())
    K = 30
    done = 0
    for k in range(K):
        one.step(a[(2 + k) % 6])
        done += int((one.gathered()[0, 0, :, -1] != 0).sum())
    many.step_many(a, K, first_block=2)
    torch.cuda.synchronize()
    assert done > 20
    assert torch.equal(many.sim.x.view(torch.int64), one.sim.x.view(torch.int64))
    assert torch.equal(many.gathered().view(torch.int64), one.gathered().view(torch.int64))
    one.close()
    many.close()


@pytest.mark.parametrize("build", ["one_wave", "two_wave"])
def test_ring_exchange_sends_a_slot_only_when_the_slowest_workgroup_has_stored_it(build, built):
    """The workgroups of a step-loop launch run at their own pace: here half of the environments fly (no contact: their
    wavefronts skip the whole constraint pipeline and run far ahead), the other half stand on the ground.  The exchange of
    step k follows the counter of step k's OWN ring slot (tds_hip_rings_t::progress[slot]), which is complete only when
    every workgroup has stored that step — every intermediate slot the communication stream copied while the launch was
    running must hold the final records (round 3 followed one running total per launch: reached by the AVERAGE
    workgroup).  shard_inplace = 0 + no communicator: the exchange of a slot is a device copy made at the moment its wait
    fires, i.e. a snapshot of what had been stored by then."""
    torch = _torch()
    m = tds_amd.load_model("ant")
    n, K = 4096, 64
    x, acts = _start(m, n, seed=3)
    x[: n // 2, 2] = 6.0  # the first half of the batch starts 6 m above the plane: airborne for the whole launch
    a = torch.from_numpy(acts).cuda().contiguous()
    opts = {"shard_inplace": 0, "exchange_w2": 1 if build == "two_wave" else 0}
    sh = hip_backend.HipShard(m, n, unique_id=None, wire_dtype="f64", options=opts)
    ref = hip_backend.HipSim(m, n, options={"exchange_w2": opts["exchange_w2"]})
    for s in (sh.sim, ref):
        s.x.copy_(torch.from_numpy(x).cuda())
    ring = torch.zeros((K, n, ref.obs_dim + 2), dtype=torch.float64, device="cuda")
    ref.step_many_rings(a, K, ring, None, first_block=2, progress=torch.zeros(K, dtype=torch.int64, device="cuda"))
    sh.step_many(a, K, first_block=2)
    sh.flush()
    torch.cuda.synchronize()
    for back in range(K):
        got = sh.gathered_step(back)
        assert torch.equal(got[0].view(torch.int64), ring[K - 1 - back].view(torch.int64)), (build, K - 1 - back)
    # (the airborne half did stay airborne, i.e. the two halves did run at different paces; a handful of the golden start
    #  states — large joint velocities, no joint limits — go non-finite in free flight within 50 steps, in the reference's
    #  own arithmetic too: they are compared bit for bit above like every other record and left out here)
    z = ref.x[: n // 2, 2]
    assert int(torch.isfinite(z).sum()) > n // 2 - 32 and float(z[torch.isfinite(z)].min()) > 1.0
    sh.close()


@pytest.mark.parametrize("with_rccl,wire,loopback,fields", [
    (False, "f32", 0, 0), (True, "f32", 0, 0), (False, "f64", 3, 0), (True, "f32", 7, 0), (False, "f32", 2, 1)])
def test_shard_peer_store_exchange_on_one_rank(with_rccl, wire, loopback, fields, built):
    """The DEFAULT ring exchange (round 5): the step-loop launch stores every step's record into the gathered slot itself —
    on every rank; here on the one rank there is, plus `loopback` scratch rings of its own standing in for peers, so that the
    kernel executes exactly what it executes on loopback + 1 GPUs — and raises the slot's flags when its last workgroup has
    stored it.  EVERY slot of the most recent launch (tds_hip_shard_gathered_step), the state and the y record, bit for bit
    against the same launches without any exchange; launches of 64 + 11 steps, three calls (both ring halves, reuse)."""
    torch = _torch()
    if with_rccl and hip_backend.HipShard.rccl_version() == 0:
        pytest.skip("librccl cannot be loaded on this machine")
    m = tds_amd.load_model("ant")
    n = 1000
    x, acts = _start(m, n, seed=43)
    a = torch.from_numpy(acts).cuda().contiguous()
    uid = hip_backend.HipShard.unique_id() if with_rccl else None
    sh = hip_backend.HipShard(m, n, unique_id=uid, wire_dtype=wire,
                              options={"shard_chunk": 64, "shard_peer": 2, "shard_peer_loopback": loopback, "exchange_fields": fields})
    ref = hip_backend.HipSim(m, n)
    for s in (sh.sim, ref):
        s.x.copy_(torch.from_numpy(x).cuda())
    K = 75
    wdt = torch.float32 if wire == "f32" else torch.float64
    idt = torch.int32 if wire == "f32" else torch.int64
    ring = torch.zeros((K, n, ref.obs_dim + 2), dtype=wdt, device="cuda")
    for rep in range(3):
        sh.step_many(a, K, first_block=1 + rep)
        assert sh.exchange_form() == "peer_stores" and sh.peer_count() == loopback
        ref.step_many_rings(a, K, ring, None, first_block=1 + rep)
        torch.cuda.synchronize()
        last = K - 64  # steps of the most recently submitted launch
        for back in range(last):
            got = sh.gathered_step(back)
            assert tuple(got.shape) == (1, n, ref.obs_dim + 2) and got.dtype == wdt
            assert torch.equal(got[0].view(idt), ring[K - 1 - back].view(idt)), (rep, back)
        assert torch.equal(sh.gathered()[0, 0].view(idt), ring[-1].view(idt))
        sh.flush()
        assert torch.equal(sh.sim.x.view(torch.int64), ref.x.view(torch.int64)), rep
        assert torch.equal(sh.sim.y.view(torch.int64), ref.y.view(torch.int64)), rep
    # an option that shapes the ring is refused once the ring exists (it would be ignored silently)
    with pytest.raises(Exception):
        sh.sim.set_option("shard_chunk", 128)
    sh.close()


@pytest.mark.parametrize("form", ["peer_stores", "peer_stores_7_loopback", "rccl_group_after_launch", "rccl_per_slot"])
def test_shard_step_many_every_gathered_slot_against_the_reference(form, built):
    """What an N > 1 run executes by default — tds_hip_shard_step_many, ring exchange — pinned on the REFERENCE
    (oracle/_ref/libtds_ref.so) at BASELINE config 3's size: every environment of every gathered slot of an Ant x 4096 x 20
    call against the reference's own step + compute_reward_done started from the state the slot before it holds (a lock-step
    plain handle provides that state: its y ring is the trajectory, and the shard's records must equal its records bit for
    bit).  One rank (a 1-GPU box); the forms differ in the kernel build and in how records become visible."""
    torch = _torch()
    from test_hip_parity import _reference_stepper
    from test_rings import _reward_done, _start_state

    m = tds_amd.load_model("ant")
    n, steps = 4096, 20
    ref_step, what = _reference_stepper("ant", n)
    rng = np.random.default_rng(78)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    opts = {"peer_stores": {"shard_peer": 2}, "peer_stores_7_loopback": {"shard_peer": 2, "shard_peer_loopback": 7},
            "rccl_group_after_launch": {"shard_peer": 0}, "rccl_per_slot": {"shard_peer": 0, "exchange_w2": 0}}[form]
    uid = hip_backend.HipShard.unique_id() if (form.startswith("rccl") and hip_backend.HipShard.rccl_version() != 0) else None
    sh = hip_backend.HipShard(m, n, unique_id=uid, wire_dtype="f64", options=opts)
    plain = hip_backend.HipSim(m, n, options={"exchange_w2": opts.get("exchange_w2", 1)})
    x0 = _start_state(m, "ant", n, rng)
    for s in (sh.sim, plain):
        s.x.copy_(torch.from_numpy(x0).cuda())
        for _ in range(10):
            s.step(None)
    act = rng.uniform(-0.4, 0.4, (steps, n, adim))
    actions = torch.from_numpy(act).cuda().contiguous()
    x_start = plain.x.cpu().numpy()
    y_ring = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
    o_ring = torch.zeros((steps, n, plain.obs_dim + 2), dtype=torch.float64, device="cuda")
    prog = torch.zeros(steps, dtype=torch.int64, device="cuda") if form == "rccl_per_slot" else None
    plain.step_many_rings(actions, steps, o_ring, y_ring, progress=prog)
    sh.step_many(actions, steps)
    assert sh.exchange_form() == form.replace("_7_loopback", "")
    slots = [sh.gathered_step(steps - 1 - k)[0].clone() for k in range(steps)]
    sh.flush()
    torch.cuda.synchronize()
    yr = y_ring.cpu().numpy()
    x = x_start.copy()
    worst = 0.0
    for k in range(steps):
        assert torch.equal(slots[k].view(torch.int64), o_ring[k].view(torch.int64)), (form, k)
        got = slots[k].cpu().numpy()
        x[:, nq + nd:nq + nd + adim] = act[k]
        y_ref = ref_step(x)
        rew, done = _reward_done(m, "ant", x[:, :nq], y_ref)
        ob = y_ref[:, :nq + nd].copy()
        ob[:, :2] = 0.0
        e = rel_err(got[:, :nq + nd], ob, floor=1e-3)
        worst = max(worst, e)
        assert e < 1e-6, (form, k, e)
        edge = np.abs(y_ref[:, 2] - 0.26) < 1e-7
        assert (got[~edge, -1] == done[~edge]).all(), (form, k)
        assert rel_err(got[~edge, -2], rew[~edge], floor=1.0) < 1e-5, (form, k)
        x[:, :nq + nd] = yr[k][:, :nq + nd]
    print(f"ant x{n}, tds_hip_shard_step_many [{form}], {steps} gathered slots, every env, vs {what}: worst {worst:.3e}")
    sh.close()
