"""worlds with SEVERAL articulated bodies (two, three, four) — each body's own forward dynamics, contacts
of each body with the plane, and contacts BETWEEN the bodies (sphere-sphere, capsule-sphere in both argument orders of
the reference's dispatcher) solved body pair by body pair, in the reference's order, with both Jacobian blocks and both
inverse mass matrices (/root/reference/src/world.hpp:206-282, 293-366; src/contact_point.hpp:43-94, 405-438, 478-495;
src/mb_constraint_solver.hpp:191-498).  The fixtures are outputs of the REAL reference (oracle/ref_harness.cpp:
RefSim::multi_body_step on data/pendulum5.urdf chains and data/sphere8cube.urdf free bodies, oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

from conftest import FLOATING_MULTI_BODY_MODELS, GOLDEN, MULTI_BODY_MODELS, rel_err

import tds_amd

TOL = 1e-6


@pytest.mark.parametrize("name", MULTI_BODY_MODELS)
def test_two_body_blob_matches_the_reference_flatten(name, built):
    """CPU: the committed blob is what include/tds_hip_stepper.hpp flattens from the reference's two MultiBody objects"""
    reflib = pytest.importorskip("reflib")
    if not reflib.available():
        pytest.skip("reference library not built here")
    import gen_golden as gen

    r, m_ref = gen.make_ref(name)
    m = tds_amd.load_model(name)
    assert tds_amd.model_to_dict(m) == tds_amd.model_to_dict(m_ref)
    nb = {"two": 2, "three": 3, "four": 4}[name.split("_")[0]]
    assert m.num_bodies == nb and [m.bodies[b].first_link for b in range(1, nb)] == [5 * b for b in range(1, nb)]
    assert m.dof_qd == 5 * nb and len(m.body_table()) == nb
    # the fixture is the reference's step on the fixture's inputs
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert rel_err(r.step(g["x"]), g["y"]) < 1e-12
    assert (g["active_contacts"] > 0).sum() >= 8      # states WITH contacts between the bodies
    assert (g["active_contacts"] == 0).sum() >= (4 if nb == 2 else 2)     # ... and without
    r.close()


@pytest.mark.parametrize("name", MULTI_BODY_MODELS)
def test_oracle_restates_two_body_worlds(name, built):
    """CPU: oracle/tds_oracle.c (step_two: per-body sub-models, pair narrowphase, two-sided MLCP) against the committed
    reference outputs and — where the reference is present — against the live reference on fresh states"""
    import oraclelib

    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert rel_err(oraclelib.step(m, g["x"]), g["y"]) < 1e-9
    reflib = pytest.importorskip("reflib")
    if reflib.available():
        import gen_golden as gen

        r, m_ref = gen.make_ref(name)
        x = gen.random_inputs(name, m_ref, 96, np.random.default_rng(777))
        assert rel_err(oraclelib.step(m, x), r.step(x)) < 1e-9
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", MULTI_BODY_MODELS)
def test_two_body_fresh_states_against_the_oracle(name, built):
    """2048 fresh states (touching and separated chains, as the fixture generator draws them) against the C oracle"""
    import torch
    import oraclelib
    from tds_amd import hip_backend

    m = tds_amd.load_model(name)
    n, nq, nd = 2048, m.dof_q, m.dof_qd
    rng = np.random.default_rng(4321)
    bt = m.body_table()
    half = bt[0]["q"][1]
    amp = 0.25 if m.has_plane else 0.9
    x = np.zeros((n, m.input_dim))
    qa = rng.uniform(-amp, amp, (n, half))
    x[:, :half] = qa
    for b in bt[1:]:   # the other chains close to the first: their spheres / capsules overlap; every fourth state apart
        x[:, b["q"][0]:b["q"][1]] = qa + rng.uniform(-0.12, 0.12, (n, half))
        x[3::4, b["q"][0]:b["q"][1]] = rng.uniform(-amp, amp, (len(x[3::4]), half))
    x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
    x[:, nq + nd:] = rng.uniform(-0.5, 0.5, (n, nd))
    sim = hip_backend.HipSim(m, n, dtype="f64")
    y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    y_ref = oraclelib.step(m, x)
    moved = np.abs(y_ref[:, nq:nq + nd] - x[:, nq:nq + nd]).max(axis=1) > 0.05   # (contacts act on most states)
    print(f"{name}: 2048 fresh states, max rel err vs the oracle {rel_err(y, y_ref):.3e}; {int(moved.sum())} with large velocity changes")
    assert rel_err(y, y_ref) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", MULTI_BODY_MODELS)
def test_two_body_single_steps(name, built):
    import torch
    from tds_amd import hip_backend

    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sim = hip_backend.HipSim(m, g["x"].shape[0], dtype="f64")
    y = sim.forward_zero(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    touching = g["active_contacts"] > 0
    e_free = rel_err(y[~touching], g["y"][~touching])
    e_touch = rel_err(y[touching], g["y"][touching])
    print(f"{name}: max rel err vs the reference — {int(touching.sum())} states with contacts between the bodies "
          f"{e_touch:.3e}, {int((~touching).sum())} without {e_free:.3e}")
    assert e_free < TOL and e_touch < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", MULTI_BODY_MODELS)
def test_two_body_closed_loop_trajectory(name, built):
    """the fixture's closed-loop trajectory (200 steps of the reference, state fed back), per-step resync"""
    import torch
    from tds_amd import hip_backend

    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    nq, nd = m.dof_q, m.dof_qd
    T = g["traj_y"].shape[0]
    x = np.tile(g["traj_x0"], (T, 1))
    x[1:, :nq + nd] = g["traj_y"][:-1, :nq + nd]            # state before step t = reference state after step t - 1
    x[:, nq + nd:nq + nd + m.action_dim] = g["traj_actions"]
    sim = hip_backend.HipSim(m, T, dtype="f64")
    y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    assert rel_err(y, g["traj_y"]) < TOL


@pytest.mark.gpu
def test_two_body_mixed_precision_and_slab(built, monkeypatch):
    """float records (double arithmetic) and the scratch-slab path (one contact's rows in LDS, the rest in the slab)"""
    import torch
    from tds_amd import hip_backend

    name = "two_pendulums_plane"
    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sim = hip_backend.HipSim(m, g["x"].shape[0], dtype="f64", na_cap=1)
    y = sim.forward_zero(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert rel_err(y, g["y"]) < TOL
    import oraclelib  # noqa: F401  (kept importable: the C oracle does not restate two-body worlds)
    x32 = g["x"].astype(np.float32)
    simf = hip_backend.HipSim(m, x32.shape[0], dtype="mixed")
    yf = simf.forward_zero(torch.from_numpy(x32).cuda()).double().cpu().numpy()
    yd = hip_backend.HipSim(m, x32.shape[0], dtype="f64").forward_zero(torch.from_numpy(x32).double().cuda()).cpu().numpy()
    assert rel_err(yf, yd) < TOL


@pytest.mark.parametrize("name", FLOATING_MULTI_BODY_MODELS)
def test_oracle_restates_multi_body_worlds_with_floating_bases(name, built):
    """CPU: free bodies among the articulated bodies of a world (two sphere8cube.urdf cubes stacked on the plane; a
    pendulum5.urdf chain and a free cube) — the oracle's sub-model per body carries the floating base through the
    single-body functions; against the committed outputs of the REAL reference (single steps, closed-loop trajectory)
    and, where it is present, the live reference"""
    import oraclelib

    m = tds_amd.load_model(name)
    assert any(b["floating"] for b in m.body_table())
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert (g["active_contacts"] > 0).sum() >= 8
    assert rel_err(oraclelib.step(m, g["x"]), g["y"]) < 1e-9
    nq, nd = m.dof_q, m.dof_qd
    T = g["traj_y"].shape[0]
    x = np.tile(g["traj_x0"], (T, 1))
    x[1:, :nq + nd] = g["traj_y"][:-1, :nq + nd]
    x[:, nq + nd:nq + nd + m.action_dim] = g["traj_actions"]
    assert rel_err(oraclelib.step(m, x), g["traj_y"]) < 1e-9
    reflib = pytest.importorskip("reflib")
    if reflib.available():
        import gen_golden as gen

        r, m_ref = gen.make_ref(name)
        assert tds_amd.model_to_dict(m) == tds_amd.model_to_dict(m_ref)
        xr = gen.random_inputs(name, m_ref, 96, np.random.default_rng(778))
        assert rel_err(oraclelib.step(m, xr), r.step(xr)) < 1e-9
        r.close()


def _floating_multi_states(name, m, n, rng):
    """states as the fixture generator draws them (oracle/gen_golden.py: random_inputs), without the reference"""
    import gen_golden as gen

    return gen.random_inputs(name, m, n, rng)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FLOATING_MULTI_BODY_MODELS)
def test_floating_multi_body_single_steps_and_trajectory(name, built):
    """GPU (kernels of KIND 4): the reference's own single steps and its 100 / 200-step closed-loop trajectory (per-step
    resync), double and float records"""
    import torch
    from tds_amd import hip_backend

    m = tds_amd.load_model(name)
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sim = hip_backend.HipSim(m, g["x"].shape[0], dtype="f64")
    y = sim.forward_zero(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    touching = g["active_contacts"] > 0
    e_free, e_touch = rel_err(y[~touching], g["y"][~touching]), rel_err(y[touching], g["y"][touching])
    print(f"{name}: max rel err vs the reference — {int(touching.sum())} states with contacts between the bodies "
          f"{e_touch:.3e}, {int((~touching).sum())} without {e_free:.3e}")
    assert e_free < TOL and e_touch < TOL
    nq, nd = m.dof_q, m.dof_qd
    T = g["traj_y"].shape[0]
    x = np.tile(g["traj_x0"], (T, 1))
    x[1:, :nq + nd] = g["traj_y"][:-1, :nq + nd]
    x[:, nq + nd:nq + nd + m.action_dim] = g["traj_actions"]
    simt = hip_backend.HipSim(m, T, dtype="f64")
    yt = simt.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    assert rel_err(yt, g["traj_y"]) < TOL
    x32 = g["x"].astype(np.float32)
    yf = hip_backend.HipSim(m, x32.shape[0], dtype="mixed").forward_zero(torch.from_numpy(x32).cuda()).double().cpu().numpy()
    yd = sim.forward_zero(torch.from_numpy(x32).double().cuda()).cpu().numpy()
    assert rel_err(yf, yd, floor=1e-6 * max(1.0, np.abs(yd).max())) < 1e-4 or rel_err(yf, yd) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", FLOATING_MULTI_BODY_MODELS)
def test_floating_multi_body_fresh_states_and_closed_loop_against_the_oracle(name, built):
    """2048 fresh states against the C oracle, then 40 closed-loop steps of all of them with per-step resync"""
    import torch
    import oraclelib
    from tds_amd import hip_backend

    m = tds_amd.load_model(name)
    n, nq, nd = 2048, m.dof_q, m.dof_qd
    x = _floating_multi_states(name, m, n, np.random.default_rng(97531))
    sim = hip_backend.HipSim(m, n, dtype="f64")
    worst = 0.0
    for t in range(40):
        y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
        y_ref = oraclelib.step(m, x)
        worst = max(worst, rel_err(y, y_ref))
        assert rel_err(y, y_ref) < TOL, (name, t)
        x[:, :nq + nd] = y_ref[:, :nq + nd]
    print(f"{name}: 2048 states x 40 closed-loop steps, worst per-step rel err vs the oracle {worst:.3e}")
