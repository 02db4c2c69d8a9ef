"""Run-time options (include/tds_hip.h: tds_hip_default_option / tds_hip_set_option / tds_hip_get_option; table:
csrc/tds_options.h): one table instead of getenv() calls scattered through the library, a snapshot per handle, nothing
latched per process."""
import numpy as np
import pytest

import tds_amd
from tds_amd import hip_backend


def test_option_table_is_exported(built):
    names = hip_backend.option_names()
    assert len(names) == len(set(names)) >= 30
    for k in ("lanes_per_env", "w2", "loop_w2", "exchange_w2", "step_many_loop", "shard_wait", "shard_inplace", "y_stride"):
        assert k in names
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.default_option("no_such_option", 1)
    hip_backend.default_option("graph_chains", 3)
    hip_backend.default_option("graph_chains", None)  # back to unset


@pytest.mark.gpu
def test_options_are_per_handle_and_the_environment_is_only_a_default(built, monkeypatch):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    m = tds_amd.load_model("ant")
    monkeypatch.setenv("TDS_HIP_STEP_MANY_LOOP", "0")
    a = hip_backend.HipSim(m, 64)                     # snapshot: step_many_loop = 0 from the environment
    monkeypatch.delenv("TDS_HIP_STEP_MANY_LOOP")
    b = hip_backend.HipSim(m, 64)                     # unset: library rule
    with hip_backend.default_options(step_many_loop=0):
        c = hip_backend.HipSim(m, 64)                 # process default beats the (absent) environment
    assert a.get_option("step_many_loop") == 0 and b.get_option("step_many_loop") is None and c.get_option("step_many_loop") == 0
    assert not a.step_many_is_loop(8) and b.step_many_is_loop(8) and not c.step_many_is_loop(8)
    a.set_option("step_many_loop", 1)                 # run-time: changes this handle only
    assert a.step_many_is_loop(8) and not c.step_many_is_loop(8)
    a.set_option("step_many_loop", None)
    assert a.get_option("step_many_loop") is None
    with pytest.raises(hip_backend.TdsHipError):      # create-time options are fixed once the handle exists
        a.set_option("w2", 0)
    with pytest.raises(hip_backend.TdsHipError):
        a.set_option("bogus", 1)
    # two handles with different step-loop builds in ONE process: same records to round-off
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "ant.npz"))
    n = 64
    x = torch.from_numpy(np.resize(g["x"], (n, m.input_dim))).cuda()
    act = torch.from_numpy(np.random.default_rng(0).uniform(-0.4, 0.4, (4, n, m.action_dim))).cuda().contiguous()
    w2 = hip_backend.HipSim(m, n)
    w1 = hip_backend.HipSim(m, n, options={"loop_w2": 0})
    ys = []
    for s in (w2, w1):
        s.x.copy_(x)
        ring = torch.zeros((6, n, m.output_dim), dtype=torch.float64, device="cuda")
        s.step_many_rings(act, 6, None, ring)
        torch.cuda.synchronize()
        ys.append(ring.cpu().numpy())
    assert __import__("conftest").rel_err(ys[0], ys[1]) < 1e-9
    assert w1.get_option("loop_w2") == 0 and w2.get_option("loop_w2") is None


@pytest.mark.gpu
def test_profile_zones_callback_reports_the_reference_zones(built):
    """tds_hip_profile_zones: the host-side counterpart of the reference's SubmitProfileTiming hook
    (/root/reference/src/base.hpp:39, world.hpp:82-86, mb_constraint_solver.hpp:225-439)"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    m = tds_amd.load_model("ant")
    sim = hip_backend.HipSim(m, 256)
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "ant.npz"))
    sim.x.copy_(torch.from_numpy(np.resize(g["x"], (256, m.input_dim))).cuda())
    z = sim.profile_zones()
    for name in ("compute multi body contacts", "solve constraints", "integrate", "inverse_mass_matrix_a", "lcpA", "solve_pgs",
                 "forward_dynamics", "step"):
        assert name in z and z[name] > 0, (name, z)
    assert z["step"] >= z["solve constraints"] + z["forward_dynamics"] and z["step"] < 1000.0


@pytest.mark.gpu
def test_an_experiment_slot_that_is_not_linked_in_is_refused(built):
    """option alt_build = k selects experiment slot k of the library (csrc/tds_kernels.h, tools/build_alt.sh); the production
    library carries none: a launch must fail loudly, never fall back to the library's own kernels"""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    m = tds_amd.load_model("ant")
    sim = hip_backend.HipSim(m, 64)
    sim.step(None)  # (the library's own kernels)
    sim.set_option("alt_build", 1)
    if hip_backend.lib_has_symbol("tds_alt_launch_1"):
        pytest.skip("this library was linked with experiment slots (tools/build_alt.sh)")
    with pytest.raises(hip_backend.TdsHipError):
        sim.step(None)
    sim.set_option("alt_build", None)
    sim.step(None)
