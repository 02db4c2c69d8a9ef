"""The serial-chain kernel (csrc/tds_chain.hip): fixed-base chains of 2 .. 8 links without contacts, torques given directly —
BASELINE configs 1 and 2 (cartpole.urdf, pendulum5.urdf; /root/reference/examples/environments/cartpole_environment.h:88-94,
src/dynamics/forward_dynamics.hpp:11-326).  Pinned on the reference's golden vectors, on the REAL reference at full size
(oracle/_ref/libtds_ref.so), on the oracle for synthetic chains of every length and joint type, and held against the general
kernel (create-time option chain = 0)."""
import ctypes as C
import os

import numpy as np
import pytest

import tds_amd
from tds_amd import hip_backend
from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-6  # BASELINE.json north_star: relative, per step
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _torch():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def _synthetic_chain(n_links, seed):
    """pendulum5's links re-used for a chain of n_links with every 1-dof joint type, tilted joint frames, joint springs and
    dampers, off-axis centres of mass and a base frame that is not the identity"""
    rng = np.random.default_rng(seed)
    m = tds_amd.load_model("pendulum5")
    J = tds_amd.model
    kinds = [J.JOINT_REVOLUTE_X, J.JOINT_REVOLUTE_Y, J.JOINT_REVOLUTE_Z, J.JOINT_REVOLUTE_AXIS, J.JOINT_PRISMATIC_X,
             J.JOINT_PRISMATIC_Y, J.JOINT_PRISMATIC_Z, J.JOINT_PRISMATIC_AXIS]
    src_link, src_vis = m.links[1], m.visuals[1]

    def rot(axis, a):
        axis = axis / np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K

    for i in range(n_links):
        if i >= 5:
            C.memmove(C.byref(m.links[i]), C.byref(src_link), C.sizeof(src_link))
            C.memmove(C.byref(m.visuals[i]), C.byref(src_vis), C.sizeof(src_vis))
        l = m.links[i]
        l.parent, l.q_index, l.qd_index = i - 1, i, i
        jt = kinds[(i + seed) % len(kinds)]
        l.joint_type = jt
        S = np.zeros(6)
        if jt in (J.JOINT_REVOLUTE_X, J.JOINT_REVOLUTE_Y, J.JOINT_REVOLUTE_Z):
            S[jt - J.JOINT_REVOLUTE_X] = 1.0
        elif jt == J.JOINT_REVOLUTE_AXIS:
            S[:3] = rng.uniform(-1, 1, 3) * 1.7  # (not normalised: link.hpp:256-261 divides by the length)
        elif jt in (J.JOINT_PRISMATIC_X, J.JOINT_PRISMATIC_Y, J.JOINT_PRISMATIC_Z):
            S[3 + jt - J.JOINT_PRISMATIC_X] = 1.0
        else:
            S[3:] = rng.uniform(-1, 1, 3)
        for k in range(6):
            l.S[k] = S[k]
        R = rot(rng.uniform(-1, 1, 3), rng.uniform(-0.8, 0.8))
        for k in range(9):
            l.X_T_rot[k] = R.flat[k]
        t = rng.uniform(-0.3, 0.3, 3) + [0, 0.4, 0]
        for k in range(3):
            l.X_T_trans[k] = t[k]
            l.com[k] = rng.uniform(-0.2, 0.4)
        l.mass = rng.uniform(0.5, 2.5)
        A = rng.uniform(-1, 1, (3, 3))
        I = A @ A.T * 0.05 + np.eye(3) * 0.1
        for k in range(9):
            l.inertia[k] = I.flat[k]
        l.stiffness = rng.uniform(0, 2.0) if i % 2 else 0.0
        l.damping = rng.uniform(0, 0.5) if i % 3 else 0.0
        m.visuals[i].link = i
        Rv = rot(rng.uniform(-1, 1, 3), rng.uniform(-3, 3))  # (large angles: every branch of matrix_to_quat)
        for k in range(9):
            m.visuals[i].X_rot[k] = Rv.flat[k]
        for k in range(3):
            m.visuals[i].X_trans[k] = rng.uniform(-0.2, 0.2)
    Rb = rot(rng.uniform(-1, 1, 3), rng.uniform(-1, 1))
    for k in range(9):
        m.base_X_world_rot[k] = Rb.flat[k]
    for k in range(3):
        m.base_X_world_trans[k] = rng.uniform(-0.5, 0.5)
    m.num_links = m.dof_q = m.dof_qd = m.action_dim = n_links
    m.num_geoms = 0
    m.num_visuals = n_links
    m.input_dim = 3 * n_links
    m.output_dim = 2 * n_links + 7 * n_links + 1
    return m


@pytest.mark.parametrize("name", ["pendulum5", "cartpole"])
@pytest.mark.parametrize("dtype", ["f64", "mixed"])
def test_golden_single_steps(name, dtype, built):
    torch = _torch()
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sim = hip_backend.HipSim(m, g["x"].shape[0], dtype=dtype)
    assert sim.single_step_kernel()[0] == "chain8"
    x = torch.from_numpy(g["x"]).to(sim.torch_dtype).cuda()
    y = sim.forward_zero(x).double().cpu().numpy()
    y_ref = g["y"]
    if dtype == "mixed":  # (the float-rounded inputs are what the kernel saw)
        import oraclelib
        y_ref = oraclelib.step(m, x.double().cpu().numpy())
    err = rel_err(y, y_ref, floor=1e-3 if dtype == "mixed" else 1e-12)
    print(f"{name} [{dtype}]: chain kernel vs reference golden, max rel err {err:.3e}")
    assert err < (TOL if dtype == "f64" else 2e-6)


def test_models_the_chain_kernel_must_not_take(built):
    _torch()
    for name in ["pendulum5_plane", "cartpole_plane", "ant", "pendulum5_spherical", "two_pendulums", "cube_floating"]:
        m = tds_amd.load_model(name)
        sim = hip_backend.HipSim(m, 8)
        assert sim.single_step_kernel()[0] != "chain8", name
    m = tds_amd.load_model("pendulum5")
    assert hip_backend.HipSim(m, 8, options={"chain": 0}).single_step_kernel()[0] == "general"
    m.visuals[1].link = 3  # (a visual on another link than its own)
    assert hip_backend.HipSim(m, 8).single_step_kernel()[0] == "general"


@pytest.mark.parametrize("n_links", [2, 3, 4, 5, 6, 7, 8])
def test_synthetic_chains_of_every_length_against_the_oracle(n_links, built):
    """every instantiation (chain length) with every joint type, tilted joint frames, springs / dampers, a tilted base: single
    steps against the oracle, 30 closed-loop steps against the general kernel, loop form == single steps"""
    torch = _torch()
    import oraclelib

    for seed in (1, 2):
        m = _synthetic_chain(n_links, seed)
        rng = np.random.default_rng(100 * n_links + seed)
        n = 67  # (ragged: the last workgroup holds three environments)
        x = np.zeros((n, m.input_dim))
        x[:, :n_links] = rng.uniform(-2.5, 2.5, (n, n_links))
        x[:, n_links:2 * n_links] = rng.uniform(-2, 2, (n, n_links))
        x[:, 2 * n_links:] = rng.uniform(-3, 3, (n, n_links))
        y_ref = oraclelib.step(m, x)
        sim = hip_backend.HipSim(m, n)
        assert sim.single_step_kernel()[0] == "chain8"
        y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
        err = rel_err(y, y_ref)
        assert err < TOL, (n_links, seed, err)
        # closed loop against the general kernel, and the loop form against single steps
        gen = hip_backend.HipSim(m, n, options={"chain": 0})
        assert gen.single_step_kernel()[0] == "general"
        steps = 30
        act = rng.uniform(-3, 3, (steps, n, n_links))
        actions = torch.from_numpy(act).cuda().contiguous()
        for s_ in (sim, gen):
            s_.x.copy_(torch.from_numpy(x).cuda())
        obs_ring = torch.full((steps, n, sim.obs_dim + 2), float("nan"), dtype=torch.float64, device="cuda")
        y_ring = torch.full((steps, n, m.output_dim), float("nan"), dtype=torch.float64, device="cuda")
        sim.step_many_rings(actions, steps, obs_ring, y_ring)
        one = hip_backend.HipSim(m, n)
        one.x.copy_(torch.from_numpy(x).cuda())
        worst = 0.0
        for k in range(steps):
            gen.step(actions[k])
            one.step(actions[k])
            assert torch.equal(one.y, y_ring[k]), (n_links, k)  # loop form == straight-line form, bit for bit
            worst = max(worst, rel_err(y_ring[k].cpu().numpy(), gen.y.cpu().numpy()))
            ob = one.x[:, :2 * n_links].clone()
            ob[:, :2] = 0
            assert torch.equal(obs_ring[k][:, :2 * n_links], ob) and (obs_ring[k][:, 2 * n_links:] == 0).all()
        assert worst < 1e-7, (n_links, seed, worst)  # (closed loop, no resync: round-off grows along the trajectory)
        print(f"chain of {n_links} links (seed {seed}): single step vs oracle {err:.2e}, {steps} closed-loop steps vs the general "
              f"kernel {worst:.2e}, loop form == single steps")


@pytest.mark.parametrize("name,n,dtype", [("pendulum5", 4096, "f64"), ("pendulum5", 4096, "mixed"), ("cartpole", 64, "f64")])
def test_closed_loop_against_the_reference_at_full_size(name, n, dtype, built):
    """BASELINE configs 2 and 1 through the form bench.py times: 50 steps as one step-loop launch with both rings on, every slot
    of every environment against the reference's own step from the state the previous slot holds"""
    torch = _torch()
    from test_hip_parity import _reference_stepper

    m = tds_amd.load_model(name)
    ref_step, what = _reference_stepper(name, n)
    rng = np.random.default_rng(5)
    nl, steps = m.num_links, 50
    sim = hip_backend.HipSim(m, n, dtype=dtype)
    assert sim.single_step_kernel()[0] == "chain8" and sim.step_many_is_loop(steps)
    tdt = sim.torch_dtype
    x0 = np.zeros((n, m.input_dim))
    x0[:, :nl] = rng.uniform(-1, 1, (n, nl))
    x0[:, nl:2 * nl] = rng.uniform(-1, 1, (n, nl))
    x0 = torch.from_numpy(x0).to(tdt).double().numpy()  # (float-representable for the float record build)
    sim.x.copy_(torch.from_numpy(x0).to(tdt).cuda())
    act = torch.from_numpy(rng.uniform(-1, 1, (steps, n, nl))).to(tdt)
    actions = act.cuda().contiguous()
    obs_ring = torch.full((steps, n, sim.obs_dim + 2), float("nan"), dtype=tdt, device="cuda")
    y_ring = torch.full((steps, n, m.output_dim), float("nan"), dtype=tdt, device="cuda")
    sim.step_many_rings(actions, steps, obs_ring, y_ring)
    yr = y_ring.double().cpu().numpy()
    assert np.isfinite(yr).all()
    # float records: the launch keeps the state in double; the reference restarts from a lock-step f64 handle's state
    state = yr
    if dtype == "mixed":
        s64 = hip_backend.HipSim(m, n, dtype="f64")
        s64.x.copy_(torch.from_numpy(x0).cuda())
        y64 = torch.zeros((steps, n, m.output_dim), dtype=torch.float64, device="cuda")
        s64.step_many_rings(actions.double().contiguous(), steps, None, y64)
        state = y64.cpu().numpy()
        assert rel_err(yr, state, floor=1e-3) < 2e-6
    tol = TOL if dtype == "f64" else 2e-6
    x = x0.copy()
    worst = 0.0
    for k in range(steps):
        x[:, 2 * nl:] = act[k].double().numpy()
        y_ref = ref_step(x)
        e = rel_err(yr[k], y_ref, floor=1e-3)
        worst = max(worst, e)
        assert e < tol, (name, k, e)
        x[:, :2 * nl] = state[k][:, :2 * nl]
    print(f"{name} x{n} [{dtype}]: {steps} ring slots of the chain kernel, every env, vs {what}: worst per-step rel err {worst:.3e}")
