"""Per-step record rings of the fast launch form (tds_hip_step_many_rings, include/tds_hip.h).

The reference hands out obs / reward / done (and the graphics state y) on EVERY step
(/root/reference/examples/ars/ars_vectorized_environment.h:240-289,
 examples/environments/locomotion_contact_simulation.h:273-303).  The step-loop launch keeps the state in LDS across its
steps; with rings it packs and stores the records of every step.  These tests pin the ring slots
  * on the REAL reference (oracle/_ref/libtds_ref.so) at BASELINE.json's full sizes, every slot of every environment;
  * on single-step launches of the same library (every kernel kind, wrap-around rings, float wire format, auto-reset).
"""
import os

import numpy as np
import pytest

import tds_amd
from tds_amd import hip_backend
from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-6  # BASELINE.json north_star: relative, per step


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _start_state(m, name, n, rng):
    """reset-like states: the environment's own reset distribution"""
    nq = m.dof_q
    x0 = np.zeros((n, m.input_dim))
    if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION:
        ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
        x0[:, 2] = 0.48
        x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
        x0[:, -3:] = [15, 0.3, 3] if name.startswith("ant") else [100, 2, 50]
    else:
        x0[:, :nq] = rng.uniform(-1, 1, (n, nq))
    return x0


def _reward_done(m, name, q_before, y):
    """compute_reward_done restated (ant_environment2.h:75-106, laikago_environment2.h:130-171)"""
    nq = m.dof_q
    q = y[:, :nq]
    if m.reward_mode == tds_amd.TDS_REWARD_ANT:
        done = q[:, 2] < 0.26
        rew = np.where(done, 0.0, (q[:, 0] - q_before[:, 0]) / m.dt)
    elif m.reward_mode == tds_amd.TDS_REWARD_LAIKAGO:
        r, p = q[:, 3], q[:, 4]
        # up_dot_world_z of quat_from_euler_rpy(roll, pitch, yaw): cos(roll) cos(pitch)
        up = np.cos(r) * np.cos(p)
        done = (up < 0.6) | (q[:, 2] < 0.2)
        rew = np.where(done, 0.0, q[:, 0])
    else:
        done = np.zeros(len(q), bool)
        rew = np.zeros(len(q))
    return rew, done


@pytest.mark.parametrize("name,n,steps,dtype,form", [
    ("ant", 4096, 20, "f64", "default"), ("pendulum5", 4096, 20, "f64", "default"), ("pendulum5", 4096, 20, "mixed", "default"),
    # config 4: the 16-lane kernel's step-loop form in wide workgroups (the default at 8192) and the single-step launches
    # of round 5 (option quad_wide = 0)
    ("laikago_soft", 8192, 50, "f64", "default"), ("laikago_soft", 8192, 20, "f64", "quad_narrow"),
    # ... in one-wavefront workgroups (resident up to 6144 environments)
    ("laikago_soft", 4096, 20, "f64", "default"), ("laikago_soft", 6144, 20, "f64", "default"),
    ("ant", 8192, 20, "f64", "default"),       # config 5's per-GPU share: the one-wave loop build
    # the builds a MULTI-GPU run launches (tds_hip_shard_step_many): a progress counter attached -> write-through record
    # stores, in the one-wave loop build (option exchange_w2 = 0) and in the two-wavefront build (the default); the obs
    # ring laid out for two ranks (in-place all-gather: obs_slot_envs); y records on 128-byte lines (y_stride)
    ("ant", 4096, 20, "f64", "exchange"), ("ant", 4096, 20, "f64", "exchange_w2"), ("ant", 4096, 20, "f64", "exchange_inplace"),
    ("ant", 4096, 20, "f64", "padded"), ("laikago_soft", 8192, 20, "f64", "padded"),
    ("ant", 4096, 20, "f64", "one_wave")])
def test_every_ring_slot_of_every_env_against_the_reference(name, n, steps, dtype, form, built):
    """BASELINE configs 3 / 2 / 4 at full size through the launch form bench.py times: K steps per call with both
    rings on; EVERY slot of EVERY environment against the reference's own step started from the state the previous
    slot holds (per-step resync costs nothing here: the y ring IS the trajectory).  `form` walks through every build of
    the step-loop kernel a run can meet — selected per HANDLE (tds_hip_set_option), several of them in one process."""
    torch = _torch()
    from test_hip_parity import _reference_stepper

    m = tds_amd.load_model(name)
    ref_step, what = _reference_stepper(name, n)
    rng = np.random.default_rng(77)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    # ("exchange": the ONE-wave build under a progress counter, round 3's default, still an option; "exchange_w2" and
    #  "exchange_inplace": the two-wavefront build — the default since round 4, what N = 1 runs)
    opts = {"exchange_w2": 1} if form == "exchange_w2" else ({"exchange_w2": 0} if form == "exchange" else
                                                               ({"loop_w2": 0} if form == "one_wave" else
                                                                ({"quad_wide": 0} if form == "quad_narrow" else None)))
    sim = hip_backend.HipSim(m, n, dtype=dtype, options=opts)
    tdt = sim.torch_dtype
    x0 = _start_state(m, name, n, rng)
    sim.x.copy_(torch.from_numpy(x0).to(tdt).cuda())
    for _ in range(10):
        sim.step(None)
    amp = 0.4 if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION else 1.0
    act = rng.uniform(-amp, amp, (steps, n, adim))
    actions = torch.from_numpy(act).to(tdt).cuda().contiguous()
    ow = sim.obs_dim + 2
    ystr = m.output_dim if form != "padded" else -(-m.output_dim // 16) * 16
    obs_envs = 2 * n if form == "exchange_inplace" else n  # (this "rank" owns the SECOND block of every slot)
    obs_ring = torch.full((steps, obs_envs, ow), float("nan"), dtype=tdt, device="cuda")
    y_ring = torch.full((steps, n, ystr), float("nan"), dtype=tdt, device="cuda")
    x_start = sim.x.cpu().numpy().astype(np.float64)
    progress = torch.zeros(steps, dtype=torch.int64, device="cuda") if form.startswith("exchange") else None  # one per slot
    if form == "exchange_inplace":
        r = sim._rings(None, y_ring, 0, 0, None)
        r.obs_ring, r.obs_slots, r.obs_first, r.obs_slot_envs = obs_ring[0, n:].data_ptr(), steps, 0, obs_envs
        r.progress = progress.data_ptr()
        sim.step_many_rings_raw(actions, steps, r)
    else:
        sim.step_many_rings(actions, steps, obs_ring, y_ring, progress=progress)
    torch.cuda.synchronize()
    if progress is not None:  # every workgroup counts itself in on the slot's own counter, once per step but the last
        assert (progress[:-1] == sim.rings_blocks()).all() and int(progress[-1].item()) == 0
    if form == "exchange_inplace":
        assert torch.isnan(obs_ring[:, :n]).all()  # (the other rank's blocks are untouched)
        obs_ring = obs_ring[:, n:]
    yr = y_ring.cpu().numpy().astype(np.float64)[:, :, :m.output_dim]
    if form == "padded":  # the padding is zero-filled: whole lines written
        assert (y_ring[:, :, m.output_dim:] == 0).all()
    orr = obs_ring.cpu().numpy().astype(np.float64)
    assert np.isfinite(yr).all() and np.isfinite(orr).all()
    assert torch.equal(sim.y, y_ring[-1][:, :m.output_dim])  # the handle's y record = the last step's
    # float records: the launch keeps the state in DOUBLE across its steps.  The reference is therefore restarted from
    # the double state of a lock-step f64 handle (same double arithmetic, double records) — not from the float-rounded
    # record of the previous slot — and every slot is held to the float contract with the ordinary floor.
    yr_state = yr
    if dtype == "mixed":
        sim64 = hip_backend.HipSim(m, n, dtype="f64", options=opts)
        sim64.x.copy_(torch.from_numpy(x_start).cuda())
        y64 = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
        sim64.step_many_rings(actions.double().contiguous(), steps, None, y64)
        torch.cuda.synchronize()
        yr_state = y64.cpu().numpy()
        assert rel_err(yr, yr_state) < 2e-6  # (the float records ARE the rounded double trajectory)
    tol = TOL if dtype == "f64" else 2e-6  # (float records: the comparison sees the rounding of inputs and outputs)
    worst = 0.0
    x = x_start.copy()
    for k in range(steps):
        a = actions[k].cpu().numpy().astype(np.float64)
        x[:, nq + nd:nq + nd + adim] = a
        y_ref = ref_step(x)
        assert np.isfinite(y_ref).all()
        floor = 1e-3
        e = rel_err(yr[k], y_ref, floor=floor)
        worst = max(worst, e)
        assert e < tol, (name, k, e)
        # the [obs | reward | done] record of the step
        rew, done = _reward_done(m, name, x[:, :nq], y_ref)
        ob = y_ref[:, :nq + nd].copy()
        ob[:, :2] = 0.0
        assert rel_err(orr[k][:, :nq + nd], ob, floor=floor) < tol, (name, k)
        if m.reward_mode != tds_amd.TDS_REWARD_NONE:
            # (an environment within round-off of a termination threshold may fall on either side)
            edge = np.zeros(n, bool)
            if m.reward_mode == tds_amd.TDS_REWARD_ANT:
                edge = np.abs(y_ref[:, 2] - 0.26) < 1e-7
            if m.reward_mode == tds_amd.TDS_REWARD_LAIKAGO:
                edge = (np.abs(np.cos(y_ref[:, 3]) * np.cos(y_ref[:, 4]) - 0.6) < 1e-7) | (np.abs(y_ref[:, 2] - 0.2) < 1e-7)
            assert (orr[k][~edge, -1] == done[~edge]).all(), (name, k)
            assert rel_err(orr[k][~edge, -2], rew[~edge], floor=1.0) < 10 * tol, (name, k)
        # resync on the device's own (double) state: what the next step started from
        x[:, :nq + nd] = yr_state[k][:, :nq + nd]
    if name == "laikago_soft":  # (step-loop form everywhere up to 8192: one-wavefront workgroups up to 6144, wide ones beyond)
        assert sim.single_step_kernel()[0] == "quad16" and sim.step_many_is_loop(steps) == (form != "quad_narrow")
    print(f"{name} x{n} [{dtype}, {form}, {sim.single_step_kernel()[0]}], {steps} ring slots, every env, vs {what}: worst "
          f"per-step rel err {worst:.3e} (loop form: {sim.step_many_is_loop(steps)})")


RING_MODELS = ["ant", "laikago", "laikago_soft", "pendulum5", "cartpole_plane", "ant_floating", "laikago_floating_env",
               "humanoid", "pendulum5_spherical", "two_pendulums_plane"]


@pytest.mark.parametrize("name", RING_MODELS)
@pytest.mark.parametrize("dtype", ["f64", "mixed"])
def test_ring_slots_equal_single_step_records(name, dtype, built):
    """every kernel kind, both forms (step-loop launch / chained graphs): slot k == what the k-th single-step launch
    of tds_hip_step_obs leaves in y and in its obs record; rings shorter than the call wrap around."""
    torch = _torch()
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    n, steps, slots = 50, 12, 5
    rng = np.random.default_rng(5)
    x = g["x"][rng.integers(0, g["x"].shape[0], n)]
    adim = m.action_dim
    sims = [hip_backend.HipSim(m, n, dtype=dtype) for _ in range(2)]
    tdt = sims[0].torch_dtype
    for s in sims:
        s.x.copy_(torch.from_numpy(x).to(tdt).cuda())
    amp = 0.4 if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION else 0.5
    actions = torch.from_numpy(rng.uniform(-amp, amp, (4, n, adim))).to(tdt).cuda().contiguous()
    obs_ring = torch.zeros((slots, n, sims[0].obs_dim + 2), dtype=tdt, device="cuda")
    y_ring = torch.zeros((steps, n, m.output_dim), dtype=tdt, device="cuda")
    sims[0].step_many_rings(actions, steps, obs_ring, y_ring, first_block=1, obs_first=3)
    obs = torch.zeros((n, sims[1].obs_dim + 2), dtype=tdt, device="cuda")
    loop = sims[0].step_many_is_loop(steps)
    # graph form: the very same kernels -> bitwise.  Step-loop launch: another compilation of the step (f64: agreement to
    # round-off, resynchronised every step); with float records it also keeps the state in double across its steps where
    # single steps round it to float once per step (tds_hip.h) — only the first step is comparable there.
    def same(a, b, k):
        if not loop:
            assert torch.equal(a, b), (name, k)
        elif dtype == "f64":
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-7, (name, k)
        elif k == 0:
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6, (name, k)
        else:
            assert torch.isfinite(a).all(), (name, k)

    nqd = sims[0].obs_dim
    for k in range(steps):
        sims[1].step(actions[(1 + k) % 4], 1, obs)
        yk, ok = y_ring[k], obs_ring[(3 + k) % slots]
        if k >= steps - slots:  # (earlier slots of the short ring have been overwritten)
            same(ok, obs, k)
        same(yk, sims[1].y, k)
        if loop and dtype == "f64":  # per-step resync on the ring's own trajectory
            sims[1].x[:, :nqd] = yk[:, :nqd]
    if not loop:
        assert torch.equal(sims[0].x, sims[1].x)
    assert torch.equal(sims[0].y, y_ring[-1])


@pytest.mark.parametrize("name", ["ant", "pendulum5"])
def test_float_obs_ring_beside_double_records_and_the_progress_counter(name, built):
    """the wire format of the multi-GPU exchange: a float obs ring written by the step-loop launch of an f64 handle, and
    the progress counters the exchange polls (per ring slot: one increment per workgroup and completed step, the last step
    excepted)"""
    torch = _torch()
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    n, steps = 203, 9
    rng = np.random.default_rng(6)
    x = g["x"][rng.integers(0, g["x"].shape[0], n)]
    sim = hip_backend.HipSim(m, n, dtype="f64")
    if not sim.step_many_is_loop(steps):
        pytest.skip("graph form")
    sim2 = hip_backend.HipSim(m, n, dtype="f64")
    for s in (sim, sim2):
        s.x.copy_(torch.from_numpy(x).cuda())
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (3, n, m.action_dim))).cuda().contiguous()
    ring32 = torch.zeros((steps, n, sim.obs_dim + 2), dtype=torch.float32, device="cuda")
    ring64 = torch.zeros((steps, n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
    progress = torch.zeros(ring32.shape[0], dtype=torch.int64, device="cuda")  # one counter per ring slot
    sim.step_many_rings(actions, steps, ring32, None, progress=progress)
    # (a progress counter selects the one-wave step-loop build; both launches get one, so that they are the same build)
    sim2.step_many_rings(actions, steps, ring64, None, progress=torch.zeros(ring64.shape[0], dtype=torch.int64, device="cuda"))
    torch.cuda.synchronize()
    assert torch.equal(ring32, ring64.to(torch.float32))
    assert int(progress.sum().item()) == (steps - 1) * sim.rings_blocks()
    assert torch.equal(sim.x, sim2.x)


@pytest.mark.parametrize("name", ["ant", "laikago"])
def test_rings_with_auto_reset(name, built):
    """auto_reset_when_done inside the ring form: slot k carries reward / done of the step that ended and the
    observation of the fresh environment (ars_vectorized_environment.h:262-289) — identical to single auto-reset steps"""
    torch = _torch()
    m = tds_amd.load_model(name)
    n, steps = 256, 40
    rng = np.random.default_rng(8)
    x0 = _start_state(m, name, n, rng)
    if name == "ant":
        x0[: n // 2, 2] = 0.27  # half of them about to end (done: z < 0.26)
    else:
        x0[: n // 2, 3] = 1.05  # ... rolled over far enough (done: up . z = cos(roll) cos(pitch) < 0.6)
    sims = [hip_backend.HipSim(m, n, dtype="f64") for _ in range(2)]
    for s in sims:
        s.x.copy_(torch.from_numpy(x0).cuda())
        s.set_auto_reset(True, 11)
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (8, n, m.action_dim))).cuda().contiguous()
    obs_ring = torch.zeros((steps, n, sims[0].obs_dim + 2), dtype=torch.float64, device="cuda")
    y_ring = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
    sims[0].step_many_rings(actions, steps, obs_ring, y_ring)
    obs = torch.zeros((n, sims[1].obs_dim + 2), dtype=torch.float64, device="cuda")
    dones = 0
    for k in range(steps):
        sims[1].step(actions[k % 8], 1, obs)
        # (step-loop launches and single steps are two compilations of the step: agreement to round-off)
        assert rel_err(obs_ring[k].cpu().numpy(), obs.cpu().numpy()) < TOL, (name, k)
        assert rel_err(y_ring[k].cpu().numpy(), sims[1].y.cpu().numpy()) < TOL, (name, k)
        dones += int((obs[:, -1] != 0).sum().item())
    assert dones >= n // 4
    assert rel_err(sims[0].x.cpu().numpy(), sims[1].x.cpu().numpy()) < TOL


@pytest.mark.parametrize("name,n", [("ant", 4096), ("laikago_soft", 4096), ("laikago_soft", 8192)])
def test_rings_with_auto_reset_against_the_reference_at_full_size(name, n, built):
    """Configs 3 and 4 through the form bench.py times with auto-reset on (config 4's default line): Ant x 4096 in the
    8-lane kernel, laikago_soft x 4096 in the 16-lane kernel and x 8192 in its wide workgroups; auto_reset_when_done, 20 steps as
    step-loop launches through the reset pool with both rings on.  The host replays the reference's loop environment by
    environment — its own step (libtds_ref.so), compute_reward_done (ant_environment2.h:75-106) and, for an environment that
    ends a step with done, reset() + the ten settle steps (ant_environment2.h:109-165; the device's counter-based random
    stream, restated in test_hip_parity._host_reset) — and every slot of both rings must agree: reward / done of the step
    that ended, the observation of the FRESH environment, the y record of the terminal state
    (ars_vectorized_environment.h:262-289)."""
    torch = _torch()
    from test_hip_parity import _host_reset, _reference_stepper

    steps, seed = 20, 23
    m = tds_amd.load_model(name)
    ref_step, what = _reference_stepper(name, n)
    rng = np.random.default_rng(99)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    x0 = _start_state(m, name, n, rng)
    sim = hip_backend.HipSim(m, n, dtype="f64")
    sim.x.copy_(torch.from_numpy(x0).cuda())
    for _ in range(10):
        sim.step(None)
    # a tenth of the batch starts just short of termination: resets in every one of the 20 steps
    xs = sim.x.cpu().numpy().copy()
    low = rng.permutation(n)[: n // 10]
    if m.reward_mode == tds_amd.TDS_REWARD_ANT:  # (just above the termination height)
        xs[low, 2] = 0.262 + 0.02 * rng.uniform(0, 1, len(low))
    else:  # Laikago: rolling over at 2 rad/s, the up vector's z just above 0.6 (laikago_environment2.h:130-171)
        xs[low, 3] = np.arccos(np.clip((0.6 + 0.02 * rng.uniform(0, 1, len(low))) / np.cos(xs[low, 4]), -1, 1))
        xs[low, nq + 3] = 2.0
    sim.x.copy_(torch.from_numpy(xs).cuda())
    sim.set_auto_reset(True, seed)
    act = rng.uniform(-0.4, 0.4, (steps, n, adim))
    actions = torch.from_numpy(act).cuda().contiguous()
    obs_ring = torch.full((steps, n, sim.obs_dim + 2), float("nan"), dtype=torch.float64, device="cuda")
    y_ring = torch.full((steps, n, m.output_dim), float("nan"), dtype=torch.float64, device="cuda")
    sim.step_many_rings(actions, steps, obs_ring, y_ring)
    torch.cuda.synchronize()
    orr, yr = obs_ring.cpu().numpy(), y_ring.cpu().numpy()
    assert np.isfinite(orr).all() and np.isfinite(yr).all()
    x = xs.copy()
    count = np.zeros(n, dtype=np.int64)  # resets of the environment so far = its position in the random stream
    resets = 0
    for k in range(steps):
        x[:, nq + nd:nq + nd + adim] = act[k]
        y_ref = ref_step(x)
        rew, done = _reward_done(m, name, x[:, :nq], y_ref)
        # (within round-off of a threshold: either side)
        if m.reward_mode == tds_amd.TDS_REWARD_ANT:
            edge = np.abs(y_ref[:, 2] - 0.26) < 1e-7
        else:
            edge = (np.abs(np.cos(y_ref[:, 3]) * np.cos(y_ref[:, 4]) - 0.6) < 1e-7) | (np.abs(y_ref[:, 2] - 0.2) < 1e-7)
        assert rel_err(yr[k], y_ref) < TOL, k
        got_done = orr[k][:, -1] != 0
        assert (got_done[~edge] == done[~edge]).all(), k
        assert rel_err(orr[k][~edge, -2], rew[~edge], floor=1.0) < 10 * TOL, k
        nxt = y_ref[:, :nq + nd].copy()
        for e in np.where(got_done)[0]:  # (the device's decision where the reference sits on the threshold)
            nxt[e] = _host_reset(m, x[e], seed, int(e), int(count[e]))
            count[e] += 1
            resets += 1
        ob = nxt.copy()
        # (ars_vectorized_environment.h:281-286: the step's observation is the state the environment goes on from — the
        #  fresh one after a reset — with the base x, y zeroed, whatever the environment's own reset() handed out)
        ob[:, :2] = 0.0
        assert rel_err(orr[k][:, :nq + nd], ob) < TOL, k
        # resync on the device's own state: what its next step started from
        x[:, :nq + nd] = np.where(got_done[:, None], orr[k][:, :nq + nd], yr[k][:, :nq + nd])
        x[got_done, :2] = nxt[got_done, :2]  # (the state keeps the base x, y)
    assert resets >= n // 12, resets
    assert rel_err(sim.x.cpu().numpy()[:, :nq + nd], x[:, :nq + nd]) < TOL
    kern = sim.single_step_kernel()[0]
    assert kern == ("oct8" if name == "ant" else "quad16") and sim.step_many_is_loop(steps)
    print(f"{name} x{n} [{kern}], auto-reset ring form, {steps} slots, {resets} resets, every env, vs {what} + host reset: ok")


@pytest.mark.gpu
def test_progress_counters_need_an_obs_ring(built):
    """round 4's advisor: the progress counters are indexed per obs-ring slot — a y ring + progress without an obs ring was a
    modulo by zero on the device; the C API refuses it (the Python binding always did)"""
    import ctypes as C

    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    m = tds_amd.load_model("ant")
    n, steps = 64, 4
    sim = hip_backend.HipSim(m, n)
    actions = torch.zeros((2, n, m.action_dim), dtype=torch.float64, device="cuda")
    y_ring = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
    progress = torch.zeros(steps, dtype=torch.int64, device="cuda")
    r = hip_backend.Rings()
    r.y_ring, r.y_slots, r.progress = y_ring.data_ptr(), steps, progress.data_ptr()
    with pytest.raises(hip_backend.TdsHipError, match="obs ring"):
        sim.step_many_rings_raw(actions, steps, r)
    assert int(progress.sum().item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cube_floating", "two_cubes_floating"])
def test_host_reset_and_host_step_of_a_model_without_actions(name, built):
    """round 4's advisor: tds_hip_reset_host staged its N-byte mask through the ACTION staging buffer — zero bytes for the
    models without actions; the staging is sized max(N x action_dim x elem, N) and allocated all-or-nothing"""
    import ctypes as C

    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    m = tds_amd.load_model(name)
    assert m.action_dim == 0
    n = 300
    sim = hip_backend.HipSim(m, n)
    L = hip_backend.lib()
    L.tds_hip_reset_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.tds_hip_step_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    # (these test worlds have no reset distribution — reset_q is all zero, a zero quaternion — so the mask selects NO
    #  environment: what is exercised is the staging of the N mask bytes and of the records; states from the golden file)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    x = g["x"][np.arange(n) % g["x"].shape[0]]
    sim.x.copy_(torch.from_numpy(x).cuda())
    x0 = sim.x.clone()
    mask = np.zeros(n, dtype=np.uint8)
    obs = np.full((n, sim.obs_dim + 2), np.nan)
    rc = L.tds_hip_reset_host(sim.h, mask.ctypes.data, obs.ctypes.data)
    assert rc == 0, hip_backend.lib().tds_hip_last_error()
    assert torch.equal(sim.x, x0)
    y = np.full((n, m.output_dim), np.nan)
    for _ in range(3):
        rc = L.tds_hip_step_host(sim.h, None, 1, obs.ctypes.data, y.ctypes.data)
        assert rc == 0, hip_backend.lib().tds_hip_last_error()
    assert np.isfinite(obs).all() and np.isfinite(y).all()
    assert not torch.equal(sim.x, x0)
