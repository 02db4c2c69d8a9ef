"""tds_hip::VectorizedEnv (include/tds_hip_stepper.hpp): the C++ class with the public surface of the reference's
VectorizedEnvironment (/root/reference/examples/ars/ars_vectorized_environment.h:141-300) and the environments resident
on the GPU.  The reference's OWN rollout loop — Worker<Env>::rollouts, examples/ars/ars_vectorized_worker.h:51-140,
compiled from the unmodified header inside oracle/_ref/libtds_ref.so — is instantiated on both classes and must walk the
same trajectories: same std::rand stream (the HIP class resets through contact_sim.reset() in this mode), same
per-environment linear policies, returns / step counts / trajectory records compared environment by environment."""
import numpy as np
import pytest

import tds_amd
from conftest import rel_err

reflib = pytest.importorskip("reflib")


def _need_harness():
    if not (reflib.available() and hasattr(reflib.lib(), "tdsref_vecenv_hip_worker")):
        pytest.skip("oracle/_ref/libtds_ref.so (with the VectorizedEnv entry points) is not built")


def test_harness_exports_the_vectorized_env_entry_points(built):
    """CPU: where the reference harness is built it carries the C++ class (compiled against the real reference headers)"""
    _need_harness()
    L = reflib.lib()
    assert hasattr(L, "tdsref_vecenv_hip_worker") and hasattr(L, "tdsref_vecenv_hip_bench")


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch,steps,auto_reset", [("ant", 24, 60, False), ("ant", 24, 120, True),
                                                         ("laikago", 12, 40, False)])
def test_reference_worker_rollouts_through_the_resident_class(name, batch, steps, auto_reset, built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _need_harness()
    m = tds_amd.load_model(name)
    od, adim = m.dof_q + m.dof_qd, m.action_dim
    rng = np.random.default_rng(5)
    # (policies strong enough to knock some Ants over: done / auto-reset come into play)
    params = rng.normal(0.0, 0.3 if name == "ant" else 0.02, (batch, adim * od + adim))
    r = reflib.vecenv_hip_worker(name, batch, steps, params, m.output_dim, shift=0.05, seed=4321, auto_reset=auto_reset)
    ref_steps, hip_steps = r["vec_steps"][0], r["vec_steps"][1]
    assert np.array_equal(r["traj_len"][0], r["traj_len"][1])
    # an environment within round-off of its termination threshold may end one step apart: everything else identical
    same = ref_steps == hip_steps
    assert same.mean() >= 0.9, (ref_steps, hip_steps)
    # closed loops of 40-120 steps through contacts amplify the per-step round-off (1e-10) of a few environments:
    # nine in ten agree to 1e-6 in their returns, all of them to 1e-3
    err = np.abs(r["total_rewards"][1] - r["total_rewards"][0]) / np.maximum(np.abs(r["total_rewards"][0]), 1e-2)
    assert (err[same] < 1e-6).mean() >= 0.9 and err[same].max() < 1e-3, err
    e_last = np.array([rel_err(r["traj_last"][1][e], r["traj_last"][0][e]) for e in range(batch)])
    # (auto-reset: an environment that ends one step apart shifts the std::rand stream of every later reset)
    assert (e_last[same] < 1e-5).mean() >= (0.75 if auto_reset else 0.9), e_last
    assert auto_reset or e_last[same].max() < 1e-2, e_last
    print(f"{name} x{batch}, {steps} steps of Worker::rollouts (auto_reset={auto_reset}): steps per env "
          f"{ref_steps.min()}..{ref_steps.max()}, returns rel err "
          f"{rel_err(r['total_rewards'][1][same], r['total_rewards'][0][same], floor=1e-2):.2e}")
    if name == "ant":
        assert (ref_steps < steps).any() or auto_reset  # (some environments did end)


@pytest.mark.gpu
@pytest.mark.parametrize("name,batch,steps,auto_reset", [("ant", 64, 150, False), ("ant", 64, 200, True),
                                                         ("laikago", 32, 60, False), ("laikago", 32, 80, True)])
def test_every_environment_of_the_resident_class_in_lock_step_with_the_reference(name, batch, steps, auto_reset, built):
    """What the rollout comparison above cannot assert (chaotic trajectories: only nine environments in ten stay within
    1e-6 over a whole rollout) holds step by step: the two classes side by side, the HIP class's state set back to the
    reference's after every step, EVERY environment's observations / rewards / y records / states within 1e-6 at every
    step and every done flag identical — host resets (auto_reset_when_done) included."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _need_harness()
    if not hasattr(reflib.lib(), "tdsref_vecenv_hip_lockstep"):
        pytest.skip("oracle/_ref/libtds_ref.so predates the lock-step entry point")
    m = tds_amd.load_model(name)
    od, adim = m.dof_q + m.dof_qd, m.action_dim
    rng = np.random.default_rng(11)
    params = rng.normal(0.0, 0.3 if name == "ant" else 0.05, (batch, adim * od + adim))
    r = reflib.vecenv_hip_lockstep(name, batch, steps, params, seed=977, auto_reset=auto_reset)
    print(f"{name} x{batch}, {steps} lock-step steps (auto_reset={auto_reset}): worst env {r['worst'].max():.2e}, median "
          f"{np.median(r['worst']):.2e}, {r['values_compared']} values, {r['host_resets']} host resets, "
          f"{r['done_mismatches']} done mismatches")
    assert r["values_compared"] >= batch * steps * (od + 1 + m.output_dim)
    assert r["worst"].max() < 1e-6, r["worst"]
    assert r["done_mismatches"] == 0
    if auto_reset and name == "ant":
        assert r["host_resets"] > 0


@pytest.mark.gpu
def test_cpp_bench_of_the_resident_class_runs(built):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    _need_harness()
    rates = reflib.vecenv_hip_bench("ant", 256, 64)
    print({k: f"{v:.3e}" for k, v in rates.items()})
    assert all(v > 0 for v in rates.values())
