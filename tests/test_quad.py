"""The 16-lane kernel of the star-shaped legged robots (csrc/tds_quad.hip; BASELINE config 4: Laikago).

A handle whose model is a root body on the reference's six virtual links + four legs of (1-dof, 1-dof, 1-dof, fixed toe) runs its
plain single steps on that kernel (four environments per wavefront, M factorised leaves-first: leg blocks, couplings, the root's
Schur complement) instead of the general 32-lane kernel.  Pinned here
  * on the reference's golden vectors (single steps, closed-loop trajectory),
  * on the general kernel (create-time option quad = 0) over states that cover 0 .. 4 penetrating toes, mixed inside a wavefront,
  * on the REAL reference (oracle/_ref/libtds_ref.so) in a closed loop of every environment at BASELINE's size,
and — because every other test of the suite that steps a Laikago model one step at a time (golden steps, ring slots of the
chained graphs, float records, the vectorised environments, the reset pool) now runs through it — by those as well."""
import os

import numpy as np
import pytest

import tds_amd
from tds_amd import hip_backend
from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
QUAD_MODELS = ["laikago", "laikago_soft"]


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


@pytest.mark.parametrize("name", QUAD_MODELS)
@pytest.mark.parametrize("dtype", ["f64", "mixed"])
def test_quad_kernel_takes_the_star_models_and_matches_the_golden_steps(name, dtype, built):
    torch = _torch()
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n = g["x"].shape[0]
    sim = hip_backend.HipSim(m, n, dtype=dtype)
    gen = hip_backend.HipSim(m, n, dtype=dtype, options={"quad": 0})
    assert sim.single_step_kernel()[:2] == ("quad16", 16) and gen.single_step_kernel()[0] == "general"
    assert hip_backend.HipSim(tds_amd.load_model("ant"), 8).single_step_kernel()[0] == "oct8"  # (its own star kernel: tests/test_oct.py)
    x = torch.from_numpy(g["x"]).to(sim.torch_dtype).cuda()
    y = sim.forward_zero(x).double().cpu().numpy()
    yg = gen.forward_zero(x).double().cpu().numpy()
    # (float records: the golden inputs are doubles — rounded to float on the way in, the step's own conditioning decides how
    #  far the double golden outputs are; what is pinned there is the general kernel on the SAME float inputs, itself held to
    #  the reference on float-representable inputs by tests/test_f32.py)
    e_ref, e_gen = rel_err(y, g["y"]), rel_err(y, yg)
    print(f"{name} [{dtype}] quad16: vs golden {e_ref:.3e}, vs the general kernel {e_gen:.3e}")
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


@pytest.mark.parametrize("name", QUAD_MODELS)
def test_quad_against_the_general_kernel_over_contact_patterns(name, built):
    """every number of penetrating toes 0 .. 4 and every mix of them inside a wavefront (the rows of a wavefront are laid out
    for its largest count): 4096 states with random base height / tilt / joint angles and velocities, one step each"""
    torch = _torch()
    m = tds_amd.load_model(name)
    n = 4096
    rng = np.random.default_rng(11)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    x = np.zeros((n, m.input_dim))
    ip = np.array([m.initial_poses[i] for i in range(adim)])
    x[:, 0:2] = rng.uniform(-2, 2, (n, 2))
    x[:, 2] = rng.uniform(0.30, 0.50, n)              # from well inside the plane to airborne
    x[:, 3:5] = rng.uniform(-0.4, 0.4, (n, 2))        # roll / pitch: some toes down, some up
    x[:, 5] = rng.uniform(-3, 3, n)
    x[:, 6:nq] = ip + rng.uniform(-0.5, 0.5, (n, nq - 6))
    x[:, nq:nq + nd] = rng.uniform(-1.5, 1.5, (n, nd))
    x[:, nq + nd:nq + nd + adim] = rng.uniform(-0.4, 0.4, (n, adim))
    x[:, -3:] = [100, 2, 50]
    sim = hip_backend.HipSim(m, n)
    gen = hip_backend.HipSim(m, n, options={"quad": 0})
    xd = torch.from_numpy(x).cuda()
    y, yg = sim.forward_zero(xd).cpu().numpy(), gen.forward_zero(xd).cpu().numpy()
    assert np.isfinite(y).all()
    e = rel_err(y, yg)
    # how many toes were down (from the oracle-free side: a toe is down when the general kernel's step changed its leg's
    # velocity differently from free flight is hard to see; count from the geometry instead: toe height at the start)
    print(f"{name}: quad16 vs general over {n} contact patterns: {e:.3e}")
    assert e < 1e-9
    # obs / reward / done records and the state feedback agree too
    o1 = torch.zeros((n, sim.obs_dim + 2), dtype=torch.float64, device="cuda")
    o2 = torch.zeros_like(o1)
    for s_, o_ in ((sim, o1), (gen, o2)):
        s_.x.copy_(xd)
        s_.step(None, 1, o_)
    assert rel_err(o1.cpu().numpy(), o2.cpu().numpy()) < 1e-9
    assert rel_err(sim.x.cpu().numpy(), gen.x.cpu().numpy()) < 1e-9
    assert (o1[:, -1] == o2[:, -1]).all()


def test_quad_closed_loop_of_every_env_against_the_reference(built):
    """BASELINE config 4 at full size through the 16-lane kernel: laikago_soft x 8192, 60 closed-loop single steps with fresh
    +-0.4 actions, every environment and every step against the REAL reference from the state the device held before the step"""
    torch = _torch()
    from test_hip_parity import _reference_stepper
    from test_rings import _start_state

    name, n, steps = "laikago_soft", 8192, 60
    m = tds_amd.load_model(name)
    ref_step, what = _reference_stepper(name, n)
    rng = np.random.default_rng(21)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    sim = hip_backend.HipSim(m, n)
    assert sim.single_step_kernel()[0] == "quad16"
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
    print(f"{name} x{n}, {steps} closed-loop steps on the 16-lane kernel, every env, vs {what}: worst per-step rel err {worst:.3e}")


@pytest.mark.parametrize("dtype", ["f64", "mixed"])
def test_quad_step_loop_form_equals_single_steps(dtype, built):
    """K steps as ONE launch of the 16-lane kernel's step-loop form (tds_hip_step_many_rings: state in LDS, a fresh action block
    per step, every step's y and obs records into ring slots that wrap around) against the same K steps as single launches
    of its straight-line form: every slot, the state and the handle's y record."""
    torch = _torch()
    name = "laikago_soft"
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n, steps, slots = 2000, 24, 7
    rng = np.random.default_rng(6)
    x = g["x"][rng.integers(0, g["x"].shape[0], n)]
    # (the library's own choice: the step-loop form while every workgroup is resident at once — up to 6144 environments —
    #  and the graphs of single steps beyond; both forced here)
    a = hip_backend.HipSim(m, n, dtype=dtype, options={"step_many_loop": 1})
    b = hip_backend.HipSim(m, n, dtype=dtype, options={"step_many_loop": 0})
    assert a.step_many_is_loop(steps) and not b.step_many_is_loop(steps) and a.single_step_kernel()[0] == "quad16"
    tdt = a.torch_dtype
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).to(tdt).cuda())
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (5, n, m.action_dim))).to(tdt).cuda().contiguous()
    obs_ring = torch.zeros((slots, n, a.obs_dim + 2), dtype=tdt, device="cuda")
    y_ring = torch.zeros((steps, n, m.output_dim), dtype=tdt, device="cuda")
    a.step_many_rings(actions, steps, obs_ring, y_ring, first_block=2, obs_first=4)
    obs = torch.zeros((n, b.obs_dim + 2), dtype=tdt, device="cuda")
    tol = 1e-9 if dtype == "f64" else 2e-6
    nqd = m.dof_q + m.dof_qd
    if dtype != "f64":
        # float records: the launch keeps the state in DOUBLE between its steps — its records are the rounded trajectory of the
        # same launch with double records (single steps would round the state to float after every step: another trajectory)
        a64 = hip_backend.HipSim(m, n, dtype="f64", options={"step_many_loop": 1})
        a64.x.copy_(a.x.double() * 0 + torch.from_numpy(x).to(tdt).cuda().double())
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
        if dtype != "f64" and k > 0:
            pass  # (no per-step record to resync on: compared loosely below)
        b.step(actions[(1 + k) % 5], 1, obs)
    lo = 1e-9 if dtype == "f64" else 1e-3
    assert rel_err(a.x.double().cpu().numpy(), b.x.double().cpu().numpy()) < lo and rel_err(o2.double().cpu().numpy(), obs.double().cpu().numpy()) < lo
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).to(tdt).cuda())
    a.step(actions[0], 4, o2)
    for k in range(4):
        b.step(actions[0], 1, obs)
    assert rel_err(a.x.double().cpu().numpy(), b.x.double().cpu().numpy()) < lo and rel_err(o2.double().cpu().numpy(), obs.double().cpu().numpy()) < lo


def test_quad_step_loop_with_auto_reset_equals_single_steps(built):
    """auto_reset_when_done inside the 16-lane kernel's step loop (a done environment takes its next pre-settled state from the
    reset pool and carries on) against single auto-reset steps through the same pool: same random stream, same records —
    a third of the environments start tilted past the termination threshold, so resets happen from the first step on."""
    torch = _torch()
    name = "laikago_soft"
    m = tds_amd.load_model(name)
    n, steps = 1024, 40
    rng = np.random.default_rng(8)
    from test_rings import _start_state

    x = _start_state(m, name, n, rng)
    x[: n // 3, 3] = rng.uniform(1.0, 1.3, n // 3)  # roll: up . z < 0.6 -> done
    a = hip_backend.HipSim(m, n)
    b = hip_backend.HipSim(m, n)
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


def test_single_step_auto_reset_in_environment_chains_equals_single_steps(built):
    """Beyond residency of the step-loop form (laikago_soft x 8192 with option quad_wide = 0: round 6's wide workgroups keep the
    loop form resident there by default) tds_hip_step_many with auto-reset runs single steps
    through the reset pool — as two ENVIRONMENT CHAINS on two streams, like the graphs of the plain call.  Against the same
    steps issued one call at a time (one whole-batch launch per step): the same records and reset stream, bit for bit; resets
    from the first step on in BOTH chains, uneven call lengths across several refill passes."""
    torch = _torch()
    name = "laikago_soft"
    m = tds_amd.load_model(name)
    n = 8192
    rng = np.random.default_rng(18)
    from test_rings import _start_state

    x = _start_state(m, name, n, rng)
    tilt = rng.permutation(n)[: n // 4]  # (spread over both halves of the batch)
    x[tilt, 3] = rng.uniform(1.0, 1.3, len(tilt))
    a = hip_backend.HipSim(m, n, options={"quad_wide": 0})
    b = hip_backend.HipSim(m, n, options={"quad_wide": 0})
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).cuda())
        s_.set_auto_reset(True, 7)
    assert not a.step_many_is_loop(8)
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (4, n, m.action_dim))).cuda().contiguous()
    obs = torch.zeros((n, b.obs_dim + 2), dtype=torch.float64, device="cuda")
    done_steps, dones = 0, 0
    for steps in (5, 33, 1, 24):
        obs_ring = torch.full((steps, n, a.obs_dim + 2), float("nan"), dtype=torch.float64, device="cuda")
        y_ring = torch.full((steps, n, m.output_dim), float("nan"), dtype=torch.float64, device="cuda")
        a.step_many_rings(actions, steps, obs_ring, y_ring, first_block=done_steps % 4)
        for k in range(steps):
            b.step(actions[(done_steps + k) % 4], 1, obs)
            dones += int((obs[:, -1] != 0).sum().item())
            assert torch.equal(obs_ring[k], obs), (done_steps, k)
            assert torch.equal(y_ring[k], b.y), (done_steps, k)
        done_steps += steps
        assert torch.equal(a.x, b.x) and torch.equal(a.y, b.y)
    half = (obs_ring[:, : n // 2, -1] != 0).sum().item(), (obs_ring[:, n // 2:, -1] != 0).sum().item()
    assert dones >= n // 4, dones
    print(f"laikago_soft x {n}: {done_steps} auto-reset steps in two environment chains == single steps, {dones} resets "
          f"(last call: {half[0]} / {half[1]} in the two halves)")


@pytest.mark.parametrize("n,auto_reset,dtype", [(8192, False, "f64"), (8192, True, "f64"), (8161, True, "f64"), (1000, False, "f64"),
                                                (8192, False, "mixed"), (8161, True, "mixed")])
def test_quad_wide_workgroups_equal_one_wavefront_workgroups_bit_for_bit(n, auto_reset, dtype, built):
    """The step-loop form in WIDE workgroups (eight wavefronts around one constant table, a workgroup per compute unit: what
    keeps laikago_soft x 8192 — config 4 — resident; option quad_wide) against the same launch in one-wavefront workgroups
    (quad_wide = 0 + step_many_loop = 1): the same kernel body, so every ring slot, the state and the reset stream bit for bit
    — with auto-reset from the first step on, a ragged last workgroup (8161 = 255 x 32 + 1) and a batch far below residency."""
    torch = _torch()
    name = "laikago_soft"
    m = tds_amd.load_model(name)
    rng = np.random.default_rng(21)
    from test_rings import _start_state

    x = _start_state(m, name, n, rng)
    if auto_reset:
        tilt = rng.permutation(n)[: n // 4]
        x[tilt, 3] = rng.uniform(1.0, 1.3, len(tilt))
    a = hip_backend.HipSim(m, n, dtype=dtype, options={"quad_wide": 2})
    b = hip_backend.HipSim(m, n, dtype=dtype, options={"quad_wide": 0, "step_many_loop": 1})
    tdt = a.torch_dtype  # (float records, double arithmetic under "mixed": the launch keeps the state in double)
    for s_ in (a, b):
        s_.x.copy_(torch.from_numpy(x).to(tdt).cuda())
        if auto_reset:
            s_.set_auto_reset(True, 11)
    assert a.step_many_is_loop(8) and b.step_many_is_loop(8)
    if n == 8192:  # the library's own choice at config 4's size
        c = hip_backend.HipSim(m, n, dtype=dtype)
        assert c.step_many_is_loop(8) and not hip_backend.HipSim(m, n, options={"quad_wide": 0}).step_many_is_loop(8)
    actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (4, n, m.action_dim))).to(tdt).cuda().contiguous()
    done_steps, dones = 0, 0
    for steps in (6, 31, 1):
        rings = []
        for s_ in (a, b):
            obs_ring = torch.full((7, n, a.obs_dim + 2), -7.0, dtype=tdt, device="cuda")
            y_ring = torch.full((steps, n, m.output_dim), -7.0, dtype=tdt, device="cuda")
            s_.step_many_rings(actions, steps, obs_ring, y_ring, first_block=done_steps % 4, obs_first=3)
            rings.append((obs_ring, y_ring))
        torch.cuda.synchronize()
        assert torch.equal(rings[0][0], rings[1][0]) and torch.equal(rings[0][1], rings[1][1])
        assert torch.isfinite(rings[0][1]).all() and (rings[0][1][:, :, -1] != -7.0).all() and (steps < 7 or (rings[0][0][:, :, 5] != -7.0).all())
        assert torch.equal(a.x, b.x) and torch.equal(a.y, b.y)
        dones += int((rings[0][0][:, :, -1] == 1.0).sum().item()) if done_steps == 0 else 0  # (the first call: 6 steps in 7 slots)
        done_steps += steps
    assert not auto_reset or dones >= n // 8, dones


def test_quad_loop_form_selection_at_the_residency_boundaries(built):
    """Which form tds_hip_step_many takes by batch size on an MI355X (256 compute units, 160 KB of LDS each): one-wavefront
    workgroups up to 6144 environments, wide workgroups up to 8192, the chained graphs beyond — and three steps through each
    side of both boundaries against single steps."""
    torch = _torch()
    name = "laikago_soft"
    m = tds_amd.load_model(name)
    rng = np.random.default_rng(31)
    from test_rings import _start_state

    props = torch.cuda.get_device_properties(0)
    if props.multi_processor_count != 256:
        pytest.skip(f"boundaries are those of 256 compute units, this device has {props.multi_processor_count}")
    for n, loop in ((6145, True), (8192, True), (8193, False)):
        a = hip_backend.HipSim(m, n)
        b = hip_backend.HipSim(m, n, options={"step_many_loop": 0})
        assert a.step_many_is_loop(3) == loop and not b.step_many_is_loop(3), n
        x = _start_state(m, name, n, rng)
        actions = torch.from_numpy(rng.uniform(-0.4, 0.4, (3, n, m.action_dim))).cuda().contiguous()
        obs = torch.zeros((n, a.obs_dim + 2), dtype=torch.float64, device="cuda")
        for s_ in (a, b):
            s_.x.copy_(torch.from_numpy(x).cuda())
        y_ring = torch.zeros((3, n, m.output_dim), dtype=torch.float64, device="cuda")
        a.step_many_rings(actions, 3, None, y_ring)
        for k in range(3):
            b.step(actions[k], 1, obs)
            assert rel_err(y_ring[k].cpu().numpy(), b.y.cpu().numpy()) < 1e-9, (n, k)
        assert rel_err(a.x.cpu().numpy(), b.x.cpu().numpy()) < 1e-9, n
