"""The 8-lane kernel of the stars with two-link legs (csrc/tds_oct.hip; BASELINE configs 3 and 5: the gym Ant).

A handle whose model is a root body on the reference's six virtual links + four legs of (hip, ankle), every leg link with a
1-dof joint, a PD actuator and a capsule, runs its plain steps on that kernel (eight environments per wavefront, M factorised
leaves-first: 2 x 2 leg blocks, couplings, the root's Schur complement; constraint rows solved and swept in windows of eight)
instead of the general 16-lane kernel.  Pinned here
  * on the reference's golden vectors (single steps, closed-loop trajectory),
  * on the general kernel (create-time option oct = 0) over states that cover 0 .. 17 penetrating contact points, mixed inside
    a wavefront, with one and with several Gauss-Seidel iterations,
  * on the REAL reference (oracle/_ref/libtds_ref.so) in a closed loop of every environment at BASELINE's sizes, single steps
    and the step-loop form's ring slots, with and without auto-reset,
and — because every other test of the suite that steps an Ant model (golden steps, ring slots, float records, the vectorised
environments, the reset pool, the C++ class) now runs through it — by those as well."""
import os

import numpy as np
import pytest

import tds_amd
from tds_amd import hip_backend
from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _contact_states(m, n, rng, lo=0.05, hi=0.75, tilt=1.2):
    """states from lying flat inside the plane (every capsule end and the torso sphere down) to airborne"""
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    x = np.zeros((n, m.input_dim))
    ip = np.array([m.initial_poses[i] for i in range(adim)])
    x[:, 0:2] = rng.uniform(-2, 2, (n, 2))
    x[:, 2] = rng.uniform(lo, hi, n)
    x[:, 3:5] = rng.uniform(-tilt, tilt, (n, 2)) * rng.uniform(0, 1, (n, 1))
    x[:, 5] = rng.uniform(-3, 3, n)
    x[:, 6:nq] = ip + rng.uniform(-0.6, 0.6, (n, nq - 6))
    x[:, nq:nq + nd] = rng.uniform(-1.5, 1.5, (n, nd))
    x[:, nq + nd:nq + nd + adim] = rng.uniform(-0.5, 0.5, (n, adim))  # (beyond the +-0.4 clamp as well)
    x[:, -3:] = [15, 0.3, 3]
    return x


@pytest.mark.parametrize("dtype", ["f64", "mixed"])
def test_oct_kernel_takes_the_ant_and_matches_the_golden_steps(dtype, built):
    torch = _torch()
    name = "ant"
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = g["x"].shape[0]
    sim = hip_backend.HipSim(m, n, dtype=dtype)
    gen = hip_backend.HipSim(m, n, dtype=dtype, options={"oct": 0})
    assert sim.single_step_kernel()[:2] == ("oct8", 8) and gen.single_step_kernel()[0] == "general"
    x = torch.from_numpy(g["x"]).to(sim.torch_dtype).cuda()
    y = sim.forward_zero(x).double().cpu().numpy()
    yg = gen.forward_zero(x).double().cpu().numpy()
    e_ref, e_gen = rel_err(y, g["y"]), rel_err(y, yg)
    print(f"{name} [{dtype}] oct8: vs golden {e_ref:.3e}, vs the general kernel {e_gen:.3e}")
    assert e_ref < (1e-6 if dtype == "f64" else 1e-3) and e_gen < (1e-9 if dtype == "f64" else 2e-6)
    # closed loop with the obs record: the golden trajectory
    x0 = np.tile(g["traj_x0"], (n, 1))
    sim.x.copy_(torch.from_numpy(x0).to(sim.torch_dtype).cuda())
    obs = torch.zeros((n, sim.obs_dim + 2), dtype=sim.torch_dtype, device="cuda")
    nqd = m.dof_q + m.dof_qd
    for t in range(30):
        a = np.tile(g["traj_actions"][t], (n, 1))
        sim.step(torch.from_numpy(a).to(sim.torch_dtype).cuda().contiguous(), 1, obs)
        assert rel_err(sim.y.double().cpu().numpy()[0], g["traj_y"][t]) < (5e-6 if dtype == "f64" else 2e-3), t
        assert torch.equal(sim.x[:, :nqd], sim.y[:, :nqd])
    assert torch.equal(obs[:, 2:nqd], sim.x[:, 2:nqd]) and (obs[:, :2] == 0).all()


def test_models_the_8_lane_kernel_must_not_take(built):
    """the kernel is built for the env step with PD control on the leg joints: a TAU-mode Ant, a PD loop that starts inside
    the root chain, the floating-base Ant and Laikago keep their kernels (the same rule now guards the
    16-lane kernel)"""
    _torch()
    ant = tds_amd.load_model("ant")
    assert hip_backend.HipSim(ant, 8).single_step_kernel()[0] == "oct8"
    m = ant.copy()
    m.pd_start_link = 5
    m.action_dim = 9
    m.input_dim = 14 + 14 + 9 + 3
    assert hip_backend.HipSim(m, 8).single_step_kernel()[0] == "general"
    assert hip_backend.HipSim(tds_amd.load_model("ant_floating"), 8).single_step_kernel()[0] == "general"
    assert hip_backend.HipSim(tds_amd.load_model("laikago"), 8).single_step_kernel()[0] == "quad16"
    lk = tds_amd.load_model("laikago").copy()
    lk.pd_start_link = 5
    lk.action_dim = 13
    lk.input_dim = 18 + 18 + 13 + 3
    assert hip_backend.HipSim(lk, 8).single_step_kernel()[0] == "general"


@pytest.mark.parametrize("iters", [1, 3])
def test_oct_against_the_general_kernel_over_contact_patterns(iters, built):
    """every number of penetrating points 0 .. 17 and every mix of them inside a wavefront (the sweep of a wavefront is laid
    out for its largest count; windows of eight rows: one to seven windows): 4096 states with random base height / tilt /
    joint angles and velocities, one step each; with pgs_iterations = 3 the windows are solved once per iteration"""
    torch = _torch()
    m = tds_amd.load_model("ant").copy()
    m.pgs_iterations = iters
    n = 4096
    rng = np.random.default_rng(11)
    x = _contact_states(m, n, rng)
    sim = hip_backend.HipSim(m, n)
    gen = hip_backend.HipSim(m, n, options={"oct": 0})
    assert sim.single_step_kernel()[0] == "oct8"
    xd = torch.from_numpy(x).cuda()
    y, yg = sim.forward_zero(xd).cpu().numpy(), gen.forward_zero(xd).cpu().numpy()
    assert np.isfinite(y).all()
    e = rel_err(y, yg)
    # contact counts, from the geometry the way the reference's narrowphase sees it (the oracle-free side: the kernel's
    # own visual poses are the links' world transforms)
    print(f"ant (pgs_iterations {iters}): oct8 vs general over {n} contact patterns: {e:.3e}")
    assert e < 1e-9
    o1 = torch.zeros((n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
    o2 = torch.zeros_like(o1)
    for s_, o_ in ((sim, o1), (gen, o2)):
        s_.x.copy_(xd)
        s_.step(None, 1, o_)
    assert rel_err(o1.cpu().numpy(), o2.cpu().numpy()) < 1e-9
    assert rel_err(sim.x.cpu().numpy(), gen.x.cpu().numpy()) < 1e-9
    assert (o1[:, -1] == o2[:, -1]).all()


def test_oct_contact_patterns_against_the_reference(built):
    """the same spread of contact patterns, 1024 states, against the REAL reference (every point of the Ant may penetrate at
    once: no row cap, no slab in this kernel)"""
    torch = _torch()
    from test_hip_parity import _reference_stepper

    m = tds_amd.load_model("ant")
    n = 1024
    ref_step, what = _reference_stepper("ant", n)
    x = _contact_states(m, n, np.random.default_rng(12))
    sim = hip_backend.HipSim(m, n)
    y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    y_ref = ref_step(x)
    e = rel_err(y, y_ref)
    print(f"ant: oct8 over {n} contact patterns vs {what}: {e:.3e}")
    assert e < 1e-6


@pytest.mark.parametrize("n", [4096, 8192])
def test_oct_closed_loop_of_every_env_against_the_reference(n, built):
    """BASELINE configs 3 / 5 (per-GPU share) through the 8-lane kernel: 60 closed-loop single steps with fresh +-0.4 actions,
    every environment and every step against the REAL reference from the state the device held before the step"""
    torch = _torch()
    from test_hip_parity import _reference_stepper
    from test_rings import _start_state

    name, steps = "ant", 60
    m = tds_amd.load_model(name)
    ref_step, what = _reference_stepper(name, n)
    rng = np.random.default_rng(21)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    sim = hip_backend.HipSim(m, n)
    assert sim.single_step_kernel()[0] == "oct8"
    sim.x.copy_(torch.from_numpy(_start_state(m, name, n, rng)).cuda())
    worst = 0.0
    for k in range(steps):
        a = rng.uniform(-0.4, 0.4, (n, adim))
        x = sim.x.cpu().numpy()
        x[:, nq + nd:nq + nd + adim] = a
        sim.step(torch.from_numpy(a).cuda().contiguous(), 1)
        y = sim.y.cpu().numpy()
        y_ref = ref_step(x)
        e = rel_err(y, y_ref)
        worst = max(worst, e)
        assert e < 1e-6, (k, e)
    print(f"{name} x{n}, {steps} closed-loop steps on the 8-lane kernel, every env, vs {what}: worst per-step rel err {worst:.3e}")


@pytest.mark.parametrize("dtype", ["f64", "mixed"])
def test_oct_step_loop_form_equals_single_steps(dtype, built):
    """K steps as ONE launch of the 8-lane kernel's step-loop form (tds_hip_step_many_rings: state in LDS, a fresh action block
    per step, every step's y and obs records into ring slots that wrap around) against the same K steps as single launches
    of its straight-line form: every slot, the state and the handle's y record.  2001 environments: a ragged last wavefront."""
    torch = _torch()
    name = "ant"
    m = tds_amd.load_model(name)
    n, steps, slots = 2001, 24, 7
    rng = np.random.default_rng(6)
    x = _contact_states(m, n, rng, lo=0.2, hi=0.6, tilt=0.5)
    a = hip_backend.HipSim(m, n, dtype=dtype, options={"step_many_loop": 1})
    b = hip_backend.HipSim(m, n, dtype=dtype, options={"step_many_loop": 0})
    assert a.step_many_is_loop(steps) and not b.step_many_is_loop(steps) and a.single_step_kernel()[0] == "oct8"
    tdt = a.torch_dtype
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).to(tdt).cuda())
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (5, n, m.action_dim))).to(tdt).cuda().contiguous()
    obs_ring = torch.zeros((slots, n, a.obs_dim + 2), dtype=tdt, device="cuda")
    y_ring = torch.zeros((steps, n, m.output_dim), dtype=tdt, device="cuda")
    a.step_many_rings(actions, steps, obs_ring, y_ring, first_block=2, obs_first=4)
    obs = torch.zeros((n, b.obs_dim + 2), dtype=tdt, device="cuda")
    tol = 1e-9 if dtype == "f64" else 2e-6
    if dtype != "f64":
        # float records: the launch keeps the state in DOUBLE between its steps — its records are the rounded trajectory of the
        # same launch with double records (single steps would round the state to float after every step: another trajectory)
        a64 = hip_backend.HipSim(m, n, dtype="f64", options={"step_many_loop": 1})
        a64.x.copy_(torch.from_numpy(x).to(tdt).cuda().double())
        y64 = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
        o64 = torch.zeros((slots, n, a.obs_dim + 2), dtype=torch.float64, device="cuda")
        a64.step_many_rings(actions.double().contiguous(), steps, o64, y64, first_block=2, obs_first=4)
        assert rel_err(y_ring.double().cpu().numpy(), y64.cpu().numpy()) < tol
        assert rel_err(obs_ring.double().cpu().numpy(), o64.cpu().numpy()) < tol
        assert torch.equal(a.y, y_ring[-1])
        return
    for k in range(steps):
        b.step(actions[(2 + k) % 5], 1, obs)
        assert rel_err(y_ring[k].double().cpu().numpy(), b.y.double().cpu().numpy()) < tol, k
        if k >= steps - slots:
            assert rel_err(obs_ring[(4 + k) % slots].double().cpu().numpy(), obs.double().cpu().numpy()) < tol, k
    assert rel_err(a.x.double().cpu().numpy(), b.x.double().cpu().numpy()) < tol
    assert torch.equal(a.y, y_ring[-1])
    # without rings: the last step's records only (tds_hip_step_many), and substeps with one action (tds_hip_step_obs)
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).to(tdt).cuda())
    o2 = torch.zeros_like(obs)
    a.step_many(actions, 9, o2, first_block=1)
    for k in range(9):
        b.step(actions[(1 + k) % 5], 1, obs)
    assert rel_err(a.x.double().cpu().numpy(), b.x.double().cpu().numpy()) < tol and rel_err(o2.double().cpu().numpy(), obs.double().cpu().numpy()) < tol
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).to(tdt).cuda())
    a.step(actions[0], 4, o2)
    for k in range(4):
        b.step(actions[0], 1, obs)
    assert rel_err(a.x.double().cpu().numpy(), b.x.double().cpu().numpy()) < tol and rel_err(o2.double().cpu().numpy(), obs.double().cpu().numpy()) < tol


def test_oct_step_loop_with_auto_reset_equals_single_steps(built):
    """auto_reset_when_done inside the 8-lane kernel's step loop (a done environment takes its next pre-settled state from the
    reset pool and carries on) against single auto-reset steps through the same pool: same random stream, same records —
    a third of the environments start below the termination height, so resets happen from the first step on."""
    torch = _torch()
    name = "ant"
    m = tds_amd.load_model(name)
    n, steps = 1024, 40
    rng = np.random.default_rng(8)
    from test_rings import _start_state

    x = _start_state(m, name, n, rng)
    x[: n // 3, 2] = rng.uniform(0.2, 0.27, n // 3)  # z < 0.26 after the step -> done
    a = hip_backend.HipSim(m, n)
    b = hip_backend.HipSim(m, n, options={"step_many_loop": 0})
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).cuda())
        s_.set_auto_reset(True, 99)
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (4, n, m.action_dim))).cuda().contiguous()
    obs_ring = torch.zeros((steps, n, a.obs_dim + 2), dtype=torch.float64, device="cuda")
    y_ring = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
    a.step_many_rings(actions, steps, obs_ring, y_ring)
    obs = torch.zeros((n, b.obs_dim + 2), dtype=torch.float64, device="cuda")
    dones = 0
    for k in range(steps):
        b.step(actions[k % 4], 1, obs)
        dones += int((obs[:, -1] != 0).sum().item())
        assert (obs_ring[k][:, -1] == obs[:, -1]).all(), k
        assert rel_err(obs_ring[k].cpu().numpy(), obs.cpu().numpy()) < 1e-9, k
        assert rel_err(y_ring[k].cpu().numpy(), b.y.cpu().numpy()) < 1e-9, k
    assert dones >= n // 3
    assert rel_err(a.x.cpu().numpy(), b.x.cpu().numpy()) < 1e-9


@pytest.mark.gpu
def test_oct_refill_passes_beside_the_chunks_equal_the_passes_behind_them(built):
    """The reset pool's refill passes of a handle whose chunks run one wavefront per SIMD use the 240-register build of the
    8-lane kernel (it fits on a SIMD beside a chunk's wavefront and is issued next to the chunk: option pool_beside, default
    on) — against the same run with the build the grid size selects (pool_beside = 0): four calls of 160 steps in chunks of
    32 with a third of the environments starting below the termination height, i.e. several passes per call and every pool
    entry consumed more than once.  Bit for bit: the kernel's builds round alike (csrc/Makefile: OCTFLAGS)."""
    torch = _torch()
    name = "ant"
    m = tds_amd.load_model(name)
    n, steps = 1024, 160
    rng = np.random.default_rng(12)
    from test_rings import _start_state

    x = _start_state(m, name, n, rng)
    x[: n // 3, 2] = rng.uniform(0.2, 0.27, n // 3)
    a = hip_backend.HipSim(m, n, options={"pool_chunk": 32})
    b = hip_backend.HipSim(m, n, options={"pool_chunk": 32, "pool_beside": 0})
    assert a.get_option("pool_beside") is None and b.get_option("pool_beside") == 0
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).cuda())
        s_.set_auto_reset(True, 5)
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (4, n, m.action_dim))).cuda().contiguous()
    rings = []
    for s_ in (a, b):
        obs_ring = torch.zeros((steps, n, s_.obs_dim + 2), dtype=torch.float64, device="cuda")
        y_ring = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
        rings.append((obs_ring, y_ring))
    dones = 0
    for call in range(4):
        for s_, (obs_ring, y_ring) in zip((a, b), rings):
            s_.step_many_rings(actions, steps, obs_ring, y_ring)
        torch.cuda.synchronize()
        assert torch.equal(rings[0][0], rings[1][0]), call
        assert torch.equal(rings[0][1], rings[1][1]), call
        assert torch.equal(a.x, b.x), call
        dones += int((rings[0][0][:, :, -1] != 0).sum().item())
    assert dones >= 4 * n  # (every environment reset several times: the rings wrapped)


def test_oct_calls_beyond_residency_run_as_environment_ranges(built):
    """A step-loop call of more environments than are resident at once in the two-wavefront build (Ant: 8192) runs as
    environment ranges one after the other, each in that build — against the same call on the one-wavefront build (option
    oct_w2 = 0: one launch, several rounds of workgroups).  Ragged on purpose: 8192 + 72 environments = a range of 8192 and one of 72.  Bit for bit (the kernel's builds round alike), every ring slot, with a different action block per step."""
    torch = _torch()
    name = "ant"
    m = tds_amd.load_model(name)
    n, steps = 8192 + 72, 24
    rng = np.random.default_rng(21)
    from test_rings import _start_state

    x = _start_state(m, name, n, rng)
    a = hip_backend.HipSim(m, n)
    b = hip_backend.HipSim(m, n, options={"oct_w2": 0})
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).cuda())
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (5, n, m.action_dim))).cuda().contiguous()
    out = []
    for s_ in (a, b):
        obs_ring = torch.zeros((steps, n, s_.obs_dim + 2), dtype=torch.float64, device="cuda")
        y_ring = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
        s_.step_many_rings(actions, steps, obs_ring, y_ring, first_block=3)
        torch.cuda.synchronize()
        out.append((obs_ring, y_ring, s_.x.clone(), s_.y.clone()))
    for u, v in zip(out[0], out[1]):
        assert torch.equal(u, v)
    assert float(out[0][0].abs().sum()) > 0 and bool(torch.isfinite(out[0][1]).all())
