"""GPU: randomised kinematic trees against the plain-C oracle — branching and chains in any link order,
massless virtual root chains of every length, fixed joints, all eight 1-dof joint types, spheres /
capsules / boxes on a tilted plane.  Exercises the general sweeps, the DPP chain hand-over (incl. the
DPP-row boundary at lane 16) and the root joint on structures no reference model has."""
import numpy as np
import pytest

from conftest import rel_err

import tds_amd
from tds_amd import hip_backend
import oraclelib

pytestmark = pytest.mark.gpu
M = tds_amd.model


def _rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def random_tree_model(seed, floating=False, spherical=False):
    """floating: the tree hangs off a floating base (random inertia, own collision shapes); links may attach
    to the base directly (parent -1) anywhere in the list, as in the reference's ant_org.urdf"""
    rng = np.random.default_rng(seed)
    m = tds_amd.Model()
    m.abi_version = tds_amd.TDS_HIP_ABI_VERSION
    m.step_mode = tds_amd.TDS_STEP_TAU
    nl = int(rng.integers(3, 23))
    n_virtual = int(rng.integers(0, 6)) if rng.random() < 0.6 else 0     # massless root chain length
    n_virtual = min(n_virtual, nl - 2)
    if floating:
        nl, n_virtual = int(rng.integers(1, 21)), 0
    if spherical:       # (a spherical joint takes three lanes and four coordinates)
        nl = int(rng.integers(2, 12))
        n_virtual = min(n_virtual, 3, nl - 1)
    nq_used = 0
    style = rng.integers(0, 3)           # 0: serial chain, 1: DFS-ordered tree, 2: arbitrary parents
    ndof = 0
    # a 6-dof virtual chain must span the motion space: prismatic x, y, z then revolute x, y, z (as URDF loaders build it)
    virt = [M.JOINT_PRISMATIC_X, M.JOINT_PRISMATIC_Y, M.JOINT_PRISMATIC_Z, M.JOINT_REVOLUTE_X, M.JOINT_REVOLUTE_Y]
    for i in range(nl):
        l = m.links[i]
        if i == 0:
            l.parent = -1
        elif i <= n_virtual or style == 0:
            l.parent = i - 1
        elif style == 1:
            l.parent = int(rng.choice([i - 1, i - 1, int(rng.integers(n_virtual, i))]))
        else:
            l.parent = int(rng.integers(-1 if floating else max(n_virtual - 1, 0), i))
        if i < n_virtual:
            jt = virt[i]
        else:
            jt = int(rng.integers(0, 8)) if (rng.random() < 0.9 or i == n_virtual) else M.JOINT_FIXED
            if spherical and (rng.random() < 0.4 or i == n_virtual) and nq_used + 4 <= 30:
                jt = M.JOINT_SPHERICAL
        l.joint_type = jt
        S = np.zeros(6)
        if jt in (M.JOINT_PRISMATIC_X, M.JOINT_PRISMATIC_Y, M.JOINT_PRISMATIC_Z):
            S[3 + jt - M.JOINT_PRISMATIC_X] = 1.0
        elif jt == M.JOINT_PRISMATIC_AXIS:
            S[3:] = rng.normal(size=3) * 0.8
        elif jt in (M.JOINT_REVOLUTE_X, M.JOINT_REVOLUTE_Y, M.JOINT_REVOLUTE_Z):
            S[jt - M.JOINT_REVOLUTE_X] = 1.0
        elif jt == M.JOINT_REVOLUTE_AXIS:
            S[:3] = rng.normal(size=3) * 0.9          # unnormalised on purpose
        for k in range(6):
            l.S[k] = S[k]
        if jt == M.JOINT_SPHERICAL:
            l.q_index, l.qd_index = nq_used, ndof
            nq_used += 4
            ndof += 3
        elif jt != M.JOINT_FIXED:
            l.q_index = (nq_used if spherical else ndof) + (7 if floating else 0)   # multi_body.hpp:324-349
            l.qd_index = ndof + (6 if floating else 0)
            nq_used += 1
            ndof += 1
        else:
            l.q_index = l.qd_index = -1
        R = _rot(rng) if i >= n_virtual else np.eye(3)
        t = rng.uniform(-0.25, 0.25, 3) if i >= n_virtual else np.zeros(3)
        for k in range(9):
            l.X_T_rot[k] = R.flat[k]
        for k in range(3):
            l.X_T_trans[k] = t[k]
        if i >= n_virtual:
            l.mass = float(rng.uniform(0.2, 2.0))
            A = rng.normal(size=(3, 3))
            I = (A @ A.T + 3 * np.eye(3)) * 0.01 * l.mass
            for k in range(9):
                l.inertia[k] = I.flat[k]
            c = rng.uniform(-0.1, 0.1, 3)
            for k in range(3):
                l.com[k] = c[k]
        l.stiffness = float(rng.uniform(0, 2)) if rng.random() < 0.2 else 0.0   # (spherical: axis-angle spring)
        l.damping = float(rng.uniform(0, 0.5)) if rng.random() < 0.2 else 0.0
    m.num_links = nl
    m.dof_q = m.dof_qd = m.action_dim = ndof
    if spherical:
        m.dof_q = nq_used
    if floating:
        m.is_floating = 1
        m.dof_q, m.dof_qd = ndof + 7, ndof + 6
        m.base_mass = float(rng.uniform(0.5, 3.0))
        A = rng.normal(size=(3, 3))
        I = (A @ A.T + 3 * np.eye(3)) * 0.01 * m.base_mass
        for k in range(9):
            m.base_inertia[k] = I.flat[k]
        for k in range(3):
            m.base_com[k] = float(rng.uniform(-0.05, 0.05))
    # contact geometry: a plane (slightly tilted) and spheres / capsules / boxes on random links
    m.has_plane = 1
    n = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), 1.0])
    n /= np.linalg.norm(n)
    for k in range(3):
        m.plane_normal[k] = n[k]
    m.plane_constant = 0.0
    ng, ncp = 0, 0
    for i in range(-1 if floating else n_virtual, nl):
        if rng.random() < 0.6 and ng < tds_amd.TDS_MAX_GEOMS:
            g = m.geoms[ng]
            kind = rng.choice(["sphere", "capsule", "box"], p=[0.6, 0.3, 0.1])
            need = {"sphere": 1, "capsule": 2, "box": 8}[kind]
            if ncp + need > 24:
                continue
            ncp += need
            g.link = i
            g.type = {"sphere": M.GEOM_SPHERE, "capsule": M.GEOM_CAPSULE, "box": M.GEOM_BOX}[kind]
            g.radius = float(rng.uniform(0.05, 0.15))
            g.length = float(rng.uniform(0.1, 0.3))
            for k in range(3):
                g.extents[k] = float(rng.uniform(0.05, 0.2))
            Rg = _rot(rng)
            tg = rng.uniform(-0.1, 0.1, 3)
            for k in range(9):
                g.X_rot[k] = Rg.flat[k]
            for k in range(3):
                g.X_trans[k] = tg[k]
            ng += 1
    m.num_geoms = ng
    m.num_visuals = 0
    m.pack_visuals = 0
    m.pgs_iterations = int(rng.integers(1, 3))
    m.input_dim = m.dof_q + m.dof_qd + ndof
    m.output_dim = m.dof_q + m.dof_qd
    m.dt = 0.005
    for k, v in enumerate((0.0, 0.0, -9.81)):
        m.gravity[k] = v
    for k in range(9):
        m.base_X_world_rot[k] = np.eye(3).flat[k]
    m.cfm, m.erp, m.friction, m.restitution = 1e-5, 0.2, float(rng.uniform(0.3, 1.0)), 0.0
    m.action_limit = 1e9
    m.name = f"rand{seed}".encode()
    return m, n_virtual, style


@pytest.mark.parametrize("seed", range(40))
def test_random_tree_against_oracle(seed, built):
    import torch
    m, n_virtual, style = random_tree_model(seed)
    rng = np.random.default_rng(1000 + seed)
    n, nd = 24, m.dof_qd
    x = np.zeros((n, m.input_dim))
    x[:, :nd] = rng.uniform(-0.6, 0.6, (n, nd))
    if n_virtual >= 3:
        x[:, 2] = rng.uniform(0.0, 0.4, n)            # height of the virtual base above the plane
    x[:, nd:2 * nd] = rng.uniform(-1, 1, (n, nd))
    x[:, 2 * nd:] = rng.uniform(-1, 1, (n, nd))
    try:
        y_ref = oraclelib.step(m, x)
    except RuntimeError:      # the reference's Cholesky inverse fails: singular joint-space inertia
        pytest.skip("degenerate random model (joint-space inertia not positive definite)")
    if not np.isfinite(y_ref).all() or np.abs(y_ref).max() > 1e6:
        pytest.skip("degenerate random model (singular joint-space inertia)")
    sim = hip_backend.HipSim(m, n)
    y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    err = rel_err(y, y_ref)
    active = int((np.array([(oraclelib.step_debug(m, x[i])["contacts"][:, 9] < 0).sum() if m.num_geoms else 0
                            for i in range(4)])).max())
    print(f"seed {seed}: {m.num_links} links ({n_virtual} virtual, style {style}), {nd} dof, "
          f"{m.num_contacts} contact points (<= {active} active), lanes {sim.kernel_info()['lanes_per_env']}: "
          f"max rel err {err:.2e}")
    assert err < 1e-6


@pytest.mark.parametrize("seed", range(30))
def test_random_floating_tree_against_oracle(seed, built):
    """floating base: random trees on a free base, any base orientation, contacts on base and links"""
    import torch
    m, _, style = random_tree_model(500 + seed, floating=True)
    rng = np.random.default_rng(2000 + seed)
    n, nq, nd, nj = 24, m.dof_q, m.dof_qd, m.action_dim
    x = np.zeros((n, m.input_dim))
    quat = rng.normal(size=(n, 4))
    x[:, 0:4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    x[:, 4:6] = rng.uniform(-1, 1, (n, 2))
    x[:, 6] = rng.uniform(0.0, 0.5, n)
    x[:, 7:nq] = rng.uniform(-0.6, 0.6, (n, nq - 7))
    x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
    x[:, nq + nd:] = rng.uniform(-1, 1, (n, nj))
    try:
        y_ref = oraclelib.step(m, x)
    except RuntimeError:
        pytest.skip("degenerate random model (joint-space inertia not positive definite)")
    if not np.isfinite(y_ref).all() or np.abs(y_ref).max() > 1e6:
        pytest.skip("degenerate random model (singular joint-space inertia)")
    sim = hip_backend.HipSim(m, n)
    y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    err = rel_err(y, y_ref)
    print(f"seed {seed}: floating base + {m.num_links} links (style {style}), {nd} dof, {m.num_contacts} contact points, "
          f"lanes {sim.kernel_info()['lanes_per_env']}: max rel err {err:.2e}")
    assert err < 1e-6
    # several steps inside one launch (the step-loop build) == the same steps one launch at a time
    xd = torch.from_numpy(x).cuda()
    sim.x.copy_(xd)
    for _ in range(3):
        sim.step(None)
    y1 = sim.y.clone()
    sim.x.copy_(xd)
    sim.step(None, 3)
    assert rel_err(sim.y.cpu().numpy(), y1.cpu().numpy()) < 1e-9


@pytest.mark.parametrize("seed", range(30))
def test_random_spherical_tree_against_oracle(seed, built):
    """spherical joints mixed with 1-dof and fixed joints in random trees, with and without a
    massless virtual root chain in front"""
    import torch
    m, n_virtual, style = random_tree_model(900 + seed, spherical=True)
    sph = [i for i in range(m.num_links) if m.links[i].joint_type == M.JOINT_SPHERICAL]
    rng = np.random.default_rng(3000 + seed)
    n, nq, nd = 24, m.dof_q, m.dof_qd
    x = np.zeros((n, m.input_dim))
    x[:, :nq] = rng.uniform(-0.6, 0.6, (n, nq))
    if n_virtual >= 3:
        x[:, 2] = rng.uniform(0.0, 0.4, n)
    for i in sph:
        quat = rng.normal(size=(n, 4))
        x[:, m.links[i].q_index:m.links[i].q_index + 4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
    x[:, nq + nd:] = rng.uniform(-1, 1, (n, nd))
    try:
        y_ref = oraclelib.step(m, x)
    except RuntimeError:
        pytest.skip("degenerate random model (joint-space inertia not positive definite)")
    if not np.isfinite(y_ref).all() or np.abs(y_ref).max() > 1e6:
        pytest.skip("degenerate random model (singular joint-space inertia)")
    sim = hip_backend.HipSim(m, n)
    y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    err = rel_err(y, y_ref)
    print(f"seed {seed}: {m.num_links} links ({len(sph)} spherical, {n_virtual} virtual, style {style}), {nd} dof, "
          f"{m.num_contacts} contact points, lanes {sim.kernel_info()['lanes_per_env']}: max rel err {err:.2e}")
    assert err < 1e-6
    xd = torch.from_numpy(x).cuda()
    sim.x.copy_(xd)
    for _ in range(3):
        sim.step(None)
    y1 = sim.y.clone()
    sim.x.copy_(xd)
    sim.step(None, 3)
    assert rel_err(sim.y.cpu().numpy(), y1.cpu().numpy()) < 1e-9


@pytest.mark.parametrize("seed", range(30))
def test_random_spherical_tree_env_step_against_oracle(seed, built):
    """the ENV step (PD controller, locomotion_contact_simulation.h:168-258) on random trees with spherical joints:
    the PD loop starts at a random link, so spherical joints lie in front of it, inside it with a link index < 4
    (visited: four pose slots, torque dropped) and inside it with an index >= 4 (euler-angle PD torque kept)"""
    import torch
    m, n_virtual, style = random_tree_model(1700 + seed, spherical=True)
    rng = np.random.default_rng(5000 + seed)
    m.step_mode = tds_amd.TDS_STEP_LOCOMOTION
    m.pd_start_link = int(rng.integers(0, min(m.num_links, 5)))
    slots = 0
    for i in range(m.pd_start_link, m.num_links):
        jt = m.links[i].joint_type
        slots += 0 if jt == M.JOINT_FIXED else 4 if jt == M.JOINT_SPHERICAL else 1
    if slots == 0 or slots > tds_amd.TDS_MAX_ACTIONS:
        pytest.skip("no PD joint / more pose slots than TDS_MAX_ACTIONS")
    m.action_dim = slots
    for k in range(slots):
        m.initial_poses[k] = float(rng.uniform(-0.3, 0.3))
    m.action_limit = 0.4
    for i in range(m.num_links):
        m.links[i].stiffness = 0.0   # (joint springs stay with the TAU-mode tests)
    n, nq, nd = 24, m.dof_q, m.dof_qd
    m.input_dim = nq + nd + slots + 3
    sph = [i for i in range(m.num_links) if m.links[i].joint_type == M.JOINT_SPHERICAL]
    x = np.zeros((n, m.input_dim))
    x[:, :nq] = rng.uniform(-0.6, 0.6, (n, nq))
    if n_virtual >= 3:
        x[:, 2] = rng.uniform(0.0, 0.4, n)
    for i in sph:
        quat = rng.normal(size=(n, 4))
        x[:, m.links[i].q_index:m.links[i].q_index + 4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
    x[:, nq + nd:nq + nd + slots] = rng.uniform(-0.6, 0.6, (n, slots))
    x[:, -3:] = [20.0, 0.5, 5.0]
    x[::2, -3:] = [60.0, 1.5, 2.0]       # the clamp to max_force is active
    try:
        y_ref = oraclelib.step(m, x)
    except RuntimeError:
        pytest.skip("degenerate random model (joint-space inertia not positive definite)")
    if not np.isfinite(y_ref).all() or np.abs(y_ref).max() > 1e6:
        pytest.skip("degenerate random model (singular joint-space inertia)")
    x0 = x.copy()
    x0[:, -3:] = 0.0
    acts = np.abs(oraclelib.step(m, x0) - y_ref).max() > 1e-6     # (False: only spherical joints below link 4 in the loop)
    sim = hip_backend.HipSim(m, n)
    y = sim.forward_zero(torch.from_numpy(x).cuda()).cpu().numpy()
    err = rel_err(y, y_ref)
    kept = [i for i in sph if i >= max(m.pd_start_link, 4)]
    print(f"seed {seed}: {m.num_links} links ({len(sph)} spherical, {len(kept)} with PD torque), PD from link "
          f"{m.pd_start_link}, {slots} pose slots, {nd} dof, controller acts: {acts}: max rel err {err:.2e}")
    assert err < 1e-6
