"""CPU, only where /root/reference exists: the plain-C oracle and the committed model blobs
against the REAL reference compiled by oracle/Makefile (oracle/_ref/libtds_ref.so)."""
import os

import numpy as np
import pytest

from conftest import MODELS, rel_err

import tds_amd
import oraclelib
import reflib

pytestmark = pytest.mark.skipif(not (os.path.isdir(reflib.REF_ROOT + "/src")),
                                reason="reference tree not present on this machine")


@pytest.fixture(scope="module")
def gen():
    import gen_golden
    return gen_golden


@pytest.mark.parametrize("name", MODELS)
def test_blob_matches_reference_flatten(name, built, gen):
    r, m_ref = gen.make_ref(name)
    m = tds_amd.load_model(name)
    assert tds_amd.model_to_dict(m) == tds_amd.model_to_dict(m_ref)
    r.close()


@pytest.mark.parametrize("name", MODELS)
def test_oracle_matches_reference_on_fresh_states(name, built, gen):
    r, m = gen.make_ref(name)
    rng = np.random.default_rng(4242)
    x = gen.random_inputs(name, m, 64, rng)
    y_ref = r.step(x)
    y = oraclelib.step(m, x)
    assert rel_err(y, y_ref) < 1e-9
    r.close()


def test_reference_intermediates(built, gen):
    r, m = gen.make_ref("ant")
    rng = np.random.default_rng(7)
    x = gen.random_inputs("ant", m, 8, rng)
    for i in range(8):
        x[i, -3:] = 0.0  # tau = 0 on both sides (reference debug path has no PD)
        dr = r.debug(x[i], m)
        do = oraclelib.step_debug(m, x[i])
        assert rel_err(do["qdd"], dr["qdd"], 1e-6) < 1e-9
        assert rel_err(do["M"], dr["M"], 1e-9) < 1e-9
        assert rel_err(do["contacts"], dr["contacts"], 1e-9) < 1e-10
        assert rel_err(do["jac"], dr["jac"], 1e-9) < 1e-10
    r.close()


def test_hipstepper_header_compiles_and_fails_loudly_without_gpu(built):
    """include/tds_hip_stepper.hpp::HipStepper inside the reference's own VectorizedEnvironment:
    compiled into oracle/_ref/libtds_ref.so; without a GPU it must raise, never fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_hip_parity.py::test_reference_vecenv_dropin")
    rc, msg, _ = reflib.hipstepper_selftest(4, 2)
    assert rc != 0 and "no HIP device" in msg


@pytest.mark.parametrize("name", ["ant", "laikago", "humanoid"])
def test_rollout_fixture_is_the_reference_worker_loop(name, built, gen):
    """tests/golden/<name>_rollout.npz is reproducible from the real reference, and the reference's loop
    (policy, step, reward/done, return bookkeeping) is what an independent numpy restatement on top of
    the plain-C oracle gives — the semantics tds_hip_rollout implements on device."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + "_rollout.npz"))
    f = gen.rollout_fixture(name)
    for k in ("x0", "params", "total_rewards", "vec_steps", "final_obs"):
        assert np.array_equal(f[k], g[k]), k
    m = tds_amd.load_model(name)
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim
    od = nq + nd
    x, params, steps, shift = g["x0"].copy(), g["params"], int(g["steps"]), float(g["shift"])
    n = x.shape[0]
    W = params[:, :adim * od].reshape(n, adim, od)   # neural_network.hpp:406-415: weights, then biases
    b = params[:, adim * od:]
    obs = x[:, :od].copy()                           # reset() hands out the raw state
    tot, cnt, done = np.zeros(n), np.zeros(n, dtype=np.int32), np.zeros(n, dtype=bool)
    for _ in range(steps):
        x[:, od:od + adim] = np.einsum("eao,eo->ea", W, obs) + b
        y = oraclelib.step(m, x)
        if name == "ant":                            # ant_environment2.h:75-106
            d = y[:, 2] < 0.26
            rew = np.where(d, 0.0, (y[:, 0] - x[:, 0]) / m.dt)
        elif name == "humanoid":                     # humanoid_environment.h:155-197
            qx, qy, qz, qw = y[:, 3], y[:, 4], y[:, 5], y[:, 6]
            up = 1.0 - 2.0 * (qx * qx + qy * qy) / (qx * qx + qy * qy + qz * qz + qw * qw)
            d = (up < 0.6) | (y[:, 2] < 0.8)
            rew = np.where(d, 0.0, y[:, 0])
        else:                                        # laikago_environment2.h:130-171
            up = np.cos(y[:, 3]) * np.cos(y[:, 4])   # R(rpy)(2,2)
            d = (up < 0.6) | (y[:, 2] < 0.2)
            rew = np.where(d, 0.0, y[:, 0])
        fresh = ~done                                # ars_vectorized_environment.h:240-289
        done = np.where(fresh, d, done)
        tot += np.where(~done, rew - shift, 0.0)
        cnt += (~done).astype(np.int32)
        x[:, :od] = y[:, :od]
        obs = y[:, :od].copy()
        obs[:, :2] = 0.0
    assert np.array_equal(cnt, g["vec_steps"])
    assert rel_err(tot, g["total_rewards"], 1e-3) < 1e-8
    assert rel_err(obs, g["final_obs"], 1e-3) < 1e-8
