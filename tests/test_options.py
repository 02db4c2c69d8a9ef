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


WATCHDOG_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["TDS_ROOT"])
import torch
import tds_amd
from tds_amd import hip_backend
name = sys.argv[1]
m = tds_amd.load_model(name)
g = np.load(os.path.join(os.environ["TDS_ROOT"], "tests", "golden", name + ".npz"))
n = 4096
x = torch.from_numpy(np.resize(g["x"], (n, m.input_dim))).cuda()
act = torch.zeros((4, n, m.action_dim), dtype=torch.float64, device="cuda")
# every step-loop launch form a caller can reach for an 8-dof model: library default, both forced compilations
for occ in (None, 2, 1):
    sim = hip_backend.HipSim(m, n)
    sim.x.copy_(x)
    refused = False
    try:
        if occ is not None:
            sim.set_option("loop_occ", occ)
        sim.step_many(act, 50)
        torch.cuda.synchronize()
    except hip_backend.TdsHipError:
        refused = True
    assert refused == (occ == 1), (name, occ, refused)
    if not refused:
        assert torch.isfinite(sim.x).all()
# ... and as the process default (the environment variable of round 4's report): refused at the launch, not hung
with hip_backend.default_options(loop_occ=1):
    sim = hip_backend.HipSim(m, n)
    sim.x.copy_(x)
    try:
        sim.step_many(act, 50)
        torch.cuda.synchronize()
        raise SystemExit("loop_occ = 1 was not refused")
    except hip_backend.TdsHipError:
        pass
print("ok")
'''


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cartpole", "pendulum5"])
def test_step_loop_launches_of_the_8_dof_kernels_terminate(name, built):
    """Round 4's advisor finding: built without MachineLICM (csrc/Makefile KFLAGS) the one-wavefront-per-SIMD compilation of
    the step loop of <double, double, 16, 8> never terminated (profiles/r04_diag_loop_hang.txt) and option loop_occ = 1 could
    still launch it.  That compilation is no longer instantiated below 14 padded dof and the option is refused there; this
    test runs every reachable step-loop form of the two 8-dof BASELINE models (configs 1, 2) under a watchdog."""
    import os
    import subprocess
    import sys

    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TDS_ROOT=root)
    p = subprocess.Popen([sys.executable, "-c", WATCHDOG_WORKER, name], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    try:
        out, _ = p.communicate(timeout=180)
    except subprocess.TimeoutExpired:
        p.kill()
        pytest.fail(f"{name}: a step-loop launch did not terminate within the watchdog's 180 s")
    assert p.returncode == 0 and b"ok" in out, out.decode(errors="replace")[-2000:]
