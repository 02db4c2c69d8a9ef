#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Generates, FROM THE REAL REFERENCE (oracle/_ref/libtds_ref.so built
from /root/reference by oracle/Makefile):

  * tiny-differentiable-simulator_amd/models/<name>.json — the flattened model blobs of the
    BASELINE.json configs (what TDS's own URDF loader + env constructors produce), and
  * tests/golden/<name>.npz — seeded input/output vectors of the reference step:
        x [N,in]  y [N,out]            single steps from randomised states (contacts on/off)
        traj_x0 [in]  traj_y [T,out]   a T-step closed-loop rollout (y[:nq+nd] fed back)
        qdd [N,nd]  M [N,nd,nd]        intermediates (ABA result, CRBA mass matrix)
  * tests/golden/<ant|laikago>_rollout.npz — the reference's Worker::rollouts loop (per-environment linear
    policy, step, reward/done, return bookkeeping) from seeded start states: x0, params, total_rewards,
    vec_steps, final_obs

Run only where /root/reference exists:   python oracle/gen_golden.py
The committed outputs are what travels to the GPU box."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import reflib  # noqa: E402
import tds_amd  # noqa: E402

# model name -> (reference constructor string, tweaks)
MODELS = {
    "cartpole": dict(ref="cartpole.urdf", dt=1e-3),                 # config 1
    "pendulum5": dict(ref="pendulum5.urdf", dt=1e-3),               # config 2
    "ant": dict(ref="ant"),                                         # config 3 / 5
    "laikago": dict(ref="laikago"),                                 # env defaults (cfm 1e-5, erp 0.2)
    "laikago_soft": dict(ref="laikago", soft=(1e4, 1e2)),           # config 4: cfm/erp from k, d
    "pendulum5_plane": dict(ref="pendulum5.urdf+plane", dt=1e-3),   # sphere contacts on a chain
    "cartpole_plane": dict(ref="cartpole.urdf+plane", dt=1e-3),     # plane-box (8 corner spheres)
    # floating base: q = [quat | pos | joints], qd = [omega | v | joints]
    "ant_floating": dict(ref="gym/ant_org.urdf+plane+floating", dt=5e-3),                   # 4 legs on the base
    "laikago_floating": dict(ref="laikago/laikago_toes_zup.urdf+plane+floating", dt=1e-3),  # 16 links, 18 dof
    "cube_floating": dict(ref="sphere8cube.urdf+plane+floating", dt=2e-3),                  # a single free body
    # the ENV step (PD controller, visual poses) on a floating base: LaikagoContactSimulation(floating = true)
    "laikago_floating_env": dict(ref="laikago_floating_env"),
    # spherical joints: 4 coordinates (quaternion) / 3 velocities per joint
    "pendulum5_spherical": dict(ref="pendulum5spherical.urdf", dt=1e-3),                          # 5 spherical links in a chain
    "sphere_spherical": dict(ref="sphere_small_xyzspherical.urdf+plane", dt=2e-3),                  # xyz prismatic + spherical
    "humanoid_spherical": dict(ref="humanoid_partial_xyz_spherical_fixed.urdf+plane", dt=1e-3),     # + fixed children, 4 shapes
    # HumanoidEnv (humanoid_environment.h): the env step on 37 links (12 of them fixed -> folded into their parents
    # on the device), xyz + spherical root joint, 21 PD-controlled joints, 27 dof
    "humanoid": dict(ref="humanoid"),
    # the PD loop's SPHERICAL branch (locomotion_contact_simulation.h:188-226; no env of the reference reaches it):
    # HumanoidContactSimulation on data/humanoid_partial_xyz_spherical.urdf with base_dof_ = 3 (the spherical root is
    # visited: pose_index += 4, torque dropped because its link index is < 4) and on data/pendulum5spherical.urdf with
    # base_dof_ = 0 (five spherical joints visited, link 4 keeps its euler-angle PD torque)
    "humanoid_sph_pd": dict(ref="humanoid_sph_pd"),
    "pendulum5_sph_pd": dict(ref="pendulum5_sph_pd"),
    # joint springs / dampers (Link::stiffness / damping; never set by the URDF loader, applied by
    # forward_dynamics.hpp:62-76 — axis-angle spring of a spherical joint — and :118-123): set on the reference's
    # links directly.  (link, stiffness, damping)
    "sphere_spherical_spring": dict(ref="sphere_small_xyzspherical.urdf+plane", dt=2e-3,
                                    springs=[(3, 0.02, 0.001), (0, 3.0, 0.2), (2, 4.0, 0.1)]),
    "pendulum5_spherical_spring": dict(ref="pendulum5spherical.urdf", dt=1e-3,
                                       springs=[(0, 1.5, 0.02), (1, 0.7, 0.0), (2, 0.0, 0.05), (4, 3.0, 0.01)]),
    # worlds with TWO articulated bodies: contacts between the links of different bodies —
    # sphere-sphere, capsule-sphere in both argument orders — on top of each body's own dynamics (and plane contacts)
    "two_pendulums": dict(ref="two:pendulum5.urdf:pendulum5.urdf:0.08:0:0", dt=1e-3),
    "two_pendulums_plane": dict(ref="two:pendulum5.urdf:pendulum5.urdf:0.08:0:0+plane", dt=1e-3),
    "two_pendulums_capsule_a": dict(ref="two:pendulum5.urdf:pendulum5.urdf:0.085:0.03:0+capsA", dt=1e-3),
    "two_pendulums_capsule_b": dict(ref="two:pendulum5.urdf:pendulum5.urdf:0.085:-0.03:0+capsB+plane", dt=1e-3),
    # worlds with THREE / FOUR articulated bodies: every pair i < j is a contact pass of its own, in the reference's
    # order (world.hpp:206-282); the chains stand in a triangle / square so that all pairs touch
    "three_pendulums": dict(ref="multi:pendulum5.urdf@0,0,0:pendulum5.urdf@0.08,0,0:pendulum5.urdf@0.04,0,0.069", dt=1e-3),
    "three_pendulums_plane": dict(ref="multi:pendulum5.urdf@0,0,0:pendulum5.urdf@0.085,0.03,0/caps:"
                                      "pendulum5.urdf@0.04,0,0.069+plane", dt=1e-3),
    "four_pendulums": dict(ref="multi:pendulum5.urdf@0,0,0:pendulum5.urdf@0.08,0,0:pendulum5.urdf@0,0,0.08:"
                               "pendulum5.urdf@0.08,0,0.08", dt=1e-3),
    # ... with FLOATING bases: two free cubes (8 corner spheres each) on the plane, one on top of the other; a fixed
    # chain and a free cube
    "two_cubes_floating": dict(ref="multi:sphere8cube.urdf@0,0,0/floating:sphere8cube.urdf@0,0,0/floating+plane", dt=2e-3),
    "pendulum_and_cube": dict(ref="multi:pendulum5.urdf@0,0,0:sphere8cube.urdf@0,0,0/floating", dt=1e-3),
}
MULTI_BODY = [n for n in MODELS if MODELS[n]["ref"].startswith(("two:", "multi:"))]


def body_table(m):
    return m.body_table()


def _spherical_links(m):
    return [i for i in range(m.num_links) if m.links[i].joint_type == tds_amd.model.JOINT_SPHERICAL]


def _set_spherical_quats(m, x, rng, spread):
    """unit quaternions (spread = size of the xyz part before normalisation) for every spherical joint"""
    for i in _spherical_links(m):
        qi = m.links[i].q_index
        quat = rng.normal(size=(x.shape[0], 4)) * [spread, spread, spread, 0.0] + [0, 0, 0, 1.0]
        x[:, qi:qi + 4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)


def make_ref(name):
    spec = MODELS[name]
    r = reflib.RefSim(spec["ref"])
    if "dt" in spec:
        r.set_dt(spec["dt"])
    for link, k, d in spec.get("springs", ()):
        r.set_link_spring(link, k, d)
    m = r.flatten()
    if "soft" in spec:
        m.set_soft_contact(*spec["soft"])
        r.set_solver(m.cfm, m.erp, m.pgs_iterations, m.friction, m.restitution)
        m = r.flatten()
    m.name = name.encode()
    return r, m


def random_inputs(name, m, n, rng):
    """Synthetic states as SURVEY.md §8(d): mirrors the envs' reset distributions, widened so
    that contacts are both active and inactive."""
    nq, nd = m.dof_q, m.dof_qd
    x = np.zeros((n, m.input_dim))
    if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION and m.is_floating:
        quat = rng.normal(size=(n, 4)) * [0.3, 0.3, 0.3, 0.0] + [0, 0, 0, 1.0]
        x[:, 0:4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
        x[:, 4:6] = rng.uniform(-1, 1, (n, 2))
        x[:, 6] = rng.uniform(0.1, 0.6, n)
        ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
        x[:, 7:nq] = ip + rng.uniform(-0.4, 0.4, (n, nq - 7))
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:nq + nd + m.action_dim] = rng.uniform(-0.6, 0.6, (n, m.action_dim))
        x[:, -3:] = [100, 2, 50]
    elif name in ("humanoid_sph_pd", "pendulum5_sph_pd"):
        x[:, :nq] = rng.uniform(-0.5, 0.5, (n, nq))
        if name == "humanoid_sph_pd":
            x[:, 2] = rng.uniform(0.1, 1.2, n)
        _set_spherical_quats(m, x, rng, 0.7 if name == "humanoid_sph_pd" else 0.06)
        if name == "pendulum5_sph_pd":
            # (the chain lies in the plane z = 0: small angles keep the penetrations shallow; the last joint — the one
            #  whose PD torque is kept — takes large rotations)
            quat = rng.normal(size=(n, 4)) * [0.8, 0.8, 0.8, 0.0] + [0, 0, 0, 1.0]
            x[:, 16:20] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
            x[::7, 16:20] = [0.0, 0.70710678, 0.0, 0.70710678]   # euler y = pi/2: the fi ~ +-1 corner of matrix_to_euler_xyz
            x[3::7, 16:20] = [0.0, -1.0, 0.0, 1.0]               # ... not normalised
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:nq + nd + m.action_dim] = rng.uniform(-0.6, 0.6, (n, m.action_dim))
        x[:, -3:] = [50, 1.5, 50]
        x[::3, -3:] = [200, 5.0, 20]             # large gains: the clamp to max_force is active
    elif name in ("two_cubes_floating", "pendulum_and_cube"):
        bt = body_table(m)
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:] = rng.uniform(-0.5, 0.5, (n, m.input_dim - nq - nd))
        for b in bt:
            if not b["floating"]:
                x[:, b["q"][0]:b["q"][1]] = rng.uniform(-0.03, 0.03, (n, b["q"][1] - b["q"][0]))
        fl = [b for b in bt if b["floating"]]
        for k, b in enumerate(fl):
            q0 = b["q"][0]
            sp = 0.06 if name == "two_cubes_floating" else 0.04
            quat = rng.normal(size=(n, 4)) * [sp, sp, sp, 0.0] + [0, 0, 0, 1.0]
            x[:, q0:q0 + 4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
            if name == "two_cubes_floating":
                # cube 0 rests on the plane (corner spheres: centre 0.4, radius 0.1), cube 1 sits on top of it, shifted a
                # little; every fourth state apart
                x[:, q0 + 4:q0 + 6] = rng.uniform(-0.06, 0.06, (n, 2))
                x[:, q0 + 6] = (0.46 + rng.uniform(0.0, 0.08, n)) if k == 0 else (0.46 + 0.93 + rng.uniform(0.0, 0.12, n))
                if k == 1:
                    x[3::4, q0 + 6] += 0.5
            else:
                # the cube hangs beside the chain (which lies along +y from the origin): a corner sphere near a link sphere
                x[:, q0 + 4] = 0.4 + rng.uniform(0.05, 0.14, n)
                x[:, q0 + 5] = 0.5 * rng.integers(1, 5, n) - 0.4 + rng.uniform(-0.03, 0.03, n)
                x[:, q0 + 6] = rng.uniform(-0.45, -0.35, n)
                x[3::4, q0 + 4] += 0.5
    elif name in MULTI_BODY and m.num_bodies > 2:
        # all chains at similar angles: their spheres / capsules overlap pair by pair; every fourth state independent
        bt = body_table(m)
        amp = 0.25 if m.has_plane else 0.9
        nj = bt[0]["q"][1]
        qa = rng.uniform(-amp, amp, (n, nj))
        for b in bt:
            x[:, b["q"][0]:b["q"][1]] = qa + rng.uniform(-0.12, 0.12, (n, nj))
            x[3::4, b["q"][0]:b["q"][1]] = rng.uniform(-amp, amp, (len(x[3::4]), nj))
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:] = rng.uniform(-1, 1, (n, nd)) * 0.5
    elif name.startswith("two_"):
        # both chains at similar angles (B = A + a little): their spheres / capsules overlap; every fourth state with
        # independent angles (separated chains).  With the plane (z = 0 through the chains' axes) small angles straddle it.
        half = nq // 2
        amp = 0.25 if m.has_plane else 0.9
        qa = rng.uniform(-amp, amp, (n, half))
        x[:, :half] = qa
        x[:, half:nq] = qa + rng.uniform(-0.12, 0.12, (n, half))
        x[3::4, half:nq] = rng.uniform(-amp, amp, (len(x[3::4]), half))
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:] = rng.uniform(-1, 1, (n, nd)) * 0.5
    elif m.step_mode == tds_amd.TDS_STEP_LOCOMOTION and _spherical_links(m):
        x[:, 0:2] = rng.uniform(-1, 1, (n, 2))
        x[:, 2] = rng.uniform(0.7, 1.5, n)       # torso height: standing ... lying on the plane
        _set_spherical_quats(m, x, rng, 0.6)
        ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
        x[:, 7:nq] = ip + rng.uniform(-0.5, 0.5, (n, nq - 7))
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:nq + nd + m.action_dim] = rng.uniform(-0.6, 0.6, (n, m.action_dim))
        x[:, -3:] = [50, 1.5, 50]                # kp, kd, max_force of HumanoidContactSimulation (humanoid_environment.h:73-75)
    elif m.step_mode == tds_amd.TDS_STEP_LOCOMOTION:
        x[:, 0:2] = rng.uniform(-1, 1, (n, 2))
        x[:, 2] = rng.uniform(0.15, 0.6, n)
        x[:, 3:6] = rng.uniform(-0.6, 0.6, (n, 3))
        ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
        x[:, 6:nq] = ip + rng.uniform(-0.4, 0.4, (n, nq - 6))
        x[:, nq:nq + nd] = rng.uniform(-2, 2, (n, nd))
        x[:, nq + nd:nq + nd + m.action_dim] = rng.uniform(-0.6, 0.6, (n, m.action_dim))
        x[:, -3:] = [15, 0.3, 3] if name.startswith("ant") else [100, 2, 50]
    elif m.is_floating:
        quat = rng.normal(size=(n, 4)) * [0.35, 0.35, 0.35, 0.0] + [0, 0, 0, 1.0]
        x[:, 0:4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
        x[:, 4:6] = rng.uniform(-1, 1, (n, 2))
        x[:, 6] = rng.uniform(0.05, 0.7, n)
        x[:, 7:nq] = rng.uniform(-0.6, 0.6, (n, nq - 7))
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:] = rng.uniform(-1, 1, (n, m.action_dim))
    elif _spherical_links(m):
        x[:, :nq] = rng.uniform(-0.3, 0.3, (n, nq))
        if m.has_plane:
            x[:, 2] = rng.uniform(-0.05, 0.4, n)     # height of the xyz base: contacts both active and not
        _set_spherical_quats(m, x, rng, 0.5)
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:] = rng.uniform(-1, 1, (n, m.action_dim))
    else:
        x[:, :nq] = rng.uniform(-1, 1, (n, nq))
        x[:, nq:nq + nd] = rng.uniform(-1, 1, (n, nd))
        x[:, nq + nd:] = rng.uniform(-1, 1, (n, nd)) * (10 if "cartpole" in name else 1)
        if name == "pendulum5_plane":
            # the chain lies along +y in the plane z=0 at q=0: small angles straddle the ground
            x[:, :nq] = rng.uniform(-0.25, 0.25, (n, nq))
    return x


def rollout_start(name, m, rng):
    nq, nd = m.dof_q, m.dof_qd
    x = np.zeros(m.input_dim)
    if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION and m.is_floating:
        ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
        x[3] = 1.0
        x[6] = 0.48
        x[7:nq] = ip + 0.05 * rng.uniform(-1, 1, nq - 7)
        x[-3:] = [100, 2, 50]
    elif name in ("humanoid_sph_pd", "pendulum5_sph_pd"):
        x[:nq] = rng.uniform(-0.1, 0.1, nq)
        if name == "humanoid_sph_pd":
            x[2] = 1.0
        xx = x[None, :].copy()
        _set_spherical_quats(m, xx, rng, 0.2)
        x[:] = xx[0]
        x[-3:] = [50, 1.5, 50]
    elif m.step_mode == tds_amd.TDS_STEP_LOCOMOTION and _spherical_links(m):
        ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
        x[2] = 1.3
        x[6] = 1.0
        x[7:nq] = ip + 0.05 * rng.uniform(-1, 1, nq - 7)
        x[-3:] = [50, 1.5, 50]
    elif m.step_mode == tds_amd.TDS_STEP_LOCOMOTION:
        ip = np.array([m.initial_poses[i] for i in range(m.action_dim)])
        x[2] = 0.48
        x[6:nq] = ip + 0.05 * rng.uniform(-1, 1, nq - 6)
        x[-3:] = [15, 0.3, 3] if name.startswith("ant") else [100, 2, 50]
    elif name == "two_cubes_floating":
        x[0:7] = [0.02, -0.03, 0.01, 1.0, 0.0, 0.0, 0.52]
        x[7:14] = [-0.03, 0.02, 0.04, 1.0, 0.03, -0.02, 0.52 + 0.98]
        x[0:4] /= np.linalg.norm(x[0:4])
        x[7:11] /= np.linalg.norm(x[7:11])
    elif name == "pendulum_and_cube":
        x[:5] = [0.3, -0.2, 0.1, 0.0, 0.1]
        x[5:12] = [0.05, 0.02, -0.04, 1.0, 0.52, 1.1, -0.38]
        x[5:9] /= np.linalg.norm(x[5:9])
        x[nq + 5 + 3] = -0.8   # the cube drifts towards the chain
    elif name in MULTI_BODY and m.num_bodies > 2:
        for k, b in enumerate(body_table(m)):
            x[b["q"][0]:b["q"][1]] = np.array([0.3, -0.2, 0.1, 0.0, 0.1]) * (1.0 - 0.15 * k) + 0.02 * k
        if m.has_plane:
            x[:nq] *= 0.5
    elif name.startswith("two_"):
        half = nq // 2
        x[:half] = [0.3, -0.2, 0.1, 0.0, 0.1]
        x[half:nq] = [0.25, -0.1, 0.05, 0.1, 0.0]
        if m.has_plane:
            x[:nq] *= 0.5
    elif m.is_floating:
        quat = np.array([0.08, -0.05, 0.02, 1.0])
        x[0:4] = quat / np.linalg.norm(quat)
        x[6] = 0.55
        x[7:nq] = rng.uniform(-0.3, 0.3, nq - 7)
        x[nq:nq + 3] = rng.uniform(-0.5, 0.5, 3)
    elif _spherical_links(m):
        x[:nq] = rng.uniform(-0.2, 0.2, nq)
        if m.has_plane:
            x[2] = 0.3
        xx = x[None, :].copy()
        _set_spherical_quats(m, xx, rng, 0.3)
        x[:] = xx[0]
    else:
        x[:nq] = rng.uniform(-1, 1, nq)
        if name == "pendulum5_plane":
            x[:nq] = [0.3, -0.2, 0.1, 0.0, 0.1]
    return x


def rollout_fixture(name, n=24, steps=25, shift=0.5, seed=77):
    """Worker::rollouts of the REAL reference (reflib.rollout) from settled start states with small random
    per-environment linear policies; every 5th environment starts in a state that ends (done) early."""
    r, m = make_ref(name)
    rng = np.random.default_rng(seed)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    od = nq + nd
    x = np.stack([rollout_start(name, m, rng) for _ in range(n)])
    for _ in range(10):  # settle with zero action, as the envs' reset() does
        y = r.step(x)
        x[:, :od] = y[:, :od]
    if name == "ant":
        x[::5, 2] -= 0.12          # torso close to the 0.26 threshold
        x[::5, 3] = 0.5
    elif name == "humanoid":
        x[::5, 2] = 0.83           # torso close to the 0.8 threshold
        tilt = np.array([0.47, 0.0, 0.0, 0.883])   # up.z = 1 - 2 (x^2 + y^2) / |q|^2 = 0.56: done from the first step
        x[::10, 3:7] = tilt / np.linalg.norm(tilt)
    else:
        x[::5, 3] = 0.92           # strongly rolled chassis: up.z drops below 0.6 during the rollout
        x[::10, 3] = 1.0           # ... or is below it from the first step
    if name == "humanoid":
        # (no joint limits, no velocity clamps: a humanoid that has fallen over blows up within tens of steps under
        #  random policies, and the reference asserts on the NaN — keep the rollout short and the policies small)
        steps = 12
    params = rng.normal(0.0, 0.02 if name == "humanoid" else 0.05, (n, adim * od + adim))
    tot, cnt, fin = reflib.rollout(name, x[:, :od], params, steps, shift)
    out = dict(x0=x, params=params, steps=np.int32(steps), shift=np.float64(shift), total_rewards=tot,
               vec_steps=cnt, final_obs=fin)
    if name in ("ant", "laikago"):
        # by-products of Worker::rollouts: RunningStat of the observations, trajectories (first 6 environments kept)
        tot2, cnt2, fin2, stats, traj, tlen = reflib.rollout_ex(name, x[:, :od], params, steps, shift, m.output_dim)
        assert np.array_equal(tot2, tot) and np.array_equal(cnt2, cnt)
        out.update(obs_stats=stats, traj=traj[:6], traj_len=tlen)
    r.close()
    return out


# hidden layers for the policy network fixture: the two ReLU layers the reference's vectorised environment keeps commented
# out (ars_vectorized_environment.h:175-176), and a second network that exercises the other activations and the input bias
NN_ACT = dict(identity=-1, tanh=0, sin=1, relu=2, soft_relu=3, elu=4, sigmoid=5, softsign=6)
NETWORKS = {
    "relu_32_64": lambda od, adim: ([od, 32, 64, adim], [NN_ACT["relu"], NN_ACT["relu"], NN_ACT["identity"]],
                                    [False, True, True, True]),
    "mixed": lambda od, adim: ([od, 24, 20, 16, 12, 10, 9, adim],
                               [NN_ACT["tanh"], NN_ACT["sin"], NN_ACT["soft_relu"], NN_ACT["elu"], NN_ACT["sigmoid"],
                                NN_ACT["softsign"], NN_ACT["identity"]],
                               [True, True, False, True, True, False, True, True]),
}


def nn_num_parameters(units, bias):
    return sum(units[i - 1] * units[i] for i in range(1, len(units))) + sum(u for u, b in zip(units, bias) if b)


def rollout_nn_fixture(name="ant", n=24, steps=25, shift=0.5, seed=79):
    """Worker::rollouts of the REAL reference with a policy NETWORK per environment (reflib.rollout(network=...)): for
    each entry of NETWORKS the parameters, returns, step counts and final observations"""
    r, m = make_ref(name)
    rng = np.random.default_rng(seed)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    od = nq + nd
    x = np.stack([rollout_start(name, m, rng) for _ in range(n)])
    for _ in range(10):
        y = r.step(x)
        x[:, :od] = y[:, :od]
    x[::5, 2] -= 0.12
    x[::5, 3] = 0.5
    out = dict(x0=x, steps=np.int32(steps), shift=np.float64(shift))
    for key, mk in NETWORKS.items():
        units, acts, bias = mk(od, adim)
        params = rng.normal(0.0, 0.25, (n, nn_num_parameters(units, bias)))
        tot, cnt, fin = reflib.rollout(name, x[:, :od], params, steps, shift, network=(units, acts, bias))
        out.update({key + "_units": np.array(units, dtype=np.int32), key + "_acts": np.array(acts, dtype=np.int32),
                    key + "_bias": np.array(bias, dtype=np.int32), key + "_params": params, key + "_total_rewards": tot,
                    key + "_vec_steps": cnt, key + "_final_obs": fin})
    r.close()
    return out


def vecenv_fixture(name, n=12, steps=60, seed=91):
    """VectorizedEnvironment::step of the REAL reference, driven like its Python binding (reflib.vecenv_steps), from
    settled start states with fresh random actions each step; every 4th environment starts near its termination
    threshold so that dones and zero rewards occur."""
    r, m = make_ref(name)
    rng = np.random.default_rng(seed)
    od = m.dof_q + m.dof_qd
    x = np.stack([rollout_start(name, m, rng) for _ in range(n)])
    for _ in range(10):
        y = r.step(x)
        x[:, :od] = y[:, :od]
    if name == "ant":
        x[::4, 2] -= 0.1
        x[::4, 3] = 0.4
    else:
        x[::4, 3] = 0.9            # rolled chassis: up.z = cos(0.9) = 0.62, drops below 0.6 within the run
        x[::8, 3] = 1.0            # ... or is below it from the first step
    acts = rng.uniform(-0.4, 0.4, (steps, n, m.action_dim))
    obs, rew, done, vis = reflib.vecenv_steps(name, x[:, :od], acts, m.output_dim)
    r.close()
    return dict(x0=x[:, :od], actions=acts, obs=obs, rewards=rew, dones=done, vis=vis[::10], vis_every=np.int32(10))


def main(only=None):
    """only: names whose .npz fixture is (re)generated — default all; the model JSONs are always rewritten."""
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    mdir = os.path.join(ROOT, "tiny-differentiable-simulator_amd", "models")
    os.makedirs(mdir, exist_ok=True)
    for idx, name in enumerate(MODELS):
        r, m = make_ref(name)
        tds_amd.save_model(m, os.path.join(mdir, name + ".json"))
        if only is not None and name not in only:
            r.close()
            continue
        rng = np.random.default_rng(1000 + idx)
        n = 48
        x = random_inputs(name, m, n, rng)
        y = r.step(x)
        nq, nd = m.dof_q, m.dof_qd
        qdd = np.zeros((n, nd))
        M = np.zeros((n, nd, nd))
        ncs = np.zeros(n, dtype=np.int32)
        for i in range(n):
            if name in MULTI_BODY:
                r.step(x[i:i + 1])   # (the intermediates of the harness are single-body: count the contacts of a step)
                npl = max(1, m.num_bodies) if m.has_plane else 0   # the plane pairs come first
                ncs[i] = sum(r.last_penetrating_contacts()[npl:])   # contacts between the bodies
                continue
            d = r.debug(x[i], m)
            qdd[i], M[i] = d["qdd"], d["M"]
            ncs[i] = int((d["contacts"][:, 9] < 0).sum()) if len(d["contacts"]) else 0
        T = 200 if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION or m.has_plane else 100
        x0 = rollout_start(name, m, rng)
        xt = x0.copy()
        traj = np.zeros((T, m.output_dim))
        acts = rng.uniform(-0.4, 0.4, (T, m.action_dim))
        if name in MULTI_BODY:
            acts *= 0.5
        elif m.step_mode != tds_amd.TDS_STEP_LOCOMOTION:
            acts *= 0.0 if name.startswith("pendulum5") else (2.0 if (m.is_floating or _spherical_links(m)) else 25.0)
        for t in range(T):
            xt[nq + nd:nq + nd + m.action_dim] = acts[t]
            traj[t] = r.step(xt)[0]
            xt[:nq + nd] = traj[t, :nq + nd]
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"),
                            x=x, y=y, qdd=qdd, M=M, active_contacts=ncs,
                            traj_x0=x0, traj_actions=acts, traj_y=traj)
        print(f"{name}: links={m.num_links} dof={nd} contacts={m.num_contacts} in={m.input_dim} "
              f"out={m.output_dim} active contacts/state: min {ncs.min()} max {ncs.max()} "
              f"mean {ncs.mean():.1f}")
        r.close()
    for name in [n for n in ("ant", "laikago", "humanoid") if only is None or n + "_rollout" in only]:
        f = rollout_fixture(name)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + "_rollout.npz"), **f)
        print(f"{name}_rollout: steps taken {f['vec_steps'].min()}..{f['vec_steps'].max()} of {int(f['steps'])}, "
              f"returns {f['total_rewards'].min():.3f}..{f['total_rewards'].max():.3f}")
    if only is None or "ant_rollout_nn" in only:
        f = rollout_nn_fixture("ant")
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ant_rollout_nn.npz"), **f)
        for key in NETWORKS:
            print(f"ant_rollout_nn[{key}]: steps taken {f[key + '_vec_steps'].min()}..{f[key + '_vec_steps'].max()}, "
                  f"returns {f[key + '_total_rewards'].min():.3f}..{f[key + '_total_rewards'].max():.3f}")
    for name in [n for n in ("ant", "laikago") if only is None or n + "_vecenv" in only]:
        f = vecenv_fixture(name)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + "_vecenv.npz"), **f)
        print(f"{name}_vecenv: {f['obs'].shape[0]} steps x {f['obs'].shape[1]} envs, dones per env "
              f"{f['dones'].sum(axis=0).astype(int).tolist()}")


if __name__ == "__main__":
    main(sys.argv[1].split(",") if len(sys.argv) > 1 else None)
