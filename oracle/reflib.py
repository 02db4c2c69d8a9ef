"""TEST INFRASTRUCTURE.  ctypes access to oracle/_ref/libtds_ref.so — the REAL reference
compiled from /root/reference by oracle/Makefile (only buildable where the reference is
present).  Used by tests/ and oracle/gen_golden.py; never by the product."""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import tds_amd  # noqa: E402

REF_ROOT = os.environ.get("TDS_REFERENCE_ROOT", "/root/reference")
_LIB_PATH = os.path.join(_HERE, "_ref", "libtds_ref.so")


def available() -> bool:
    return os.path.exists(_LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (libtds_ref.so links libtds_hip.so: one HIP runtime per process)
        L = C.CDLL(_LIB_PATH)
        L.tdsref_create.restype = C.c_void_p
        L.tdsref_create.argtypes = [C.c_char_p, C.c_char_p]
        L.tdsref_destroy.argtypes = [C.c_void_p]
        L.tdsref_input_dim.argtypes = [C.c_void_p]
        L.tdsref_output_dim.argtypes = [C.c_void_p]
        L.tdsref_set_dt.argtypes = [C.c_void_p, C.c_double]
        L.tdsref_set_gravity.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double]
        L.tdsref_set_solver.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_double,
                                        C.c_double]
        if hasattr(L, "tdsref_set_link_spring"):
            L.tdsref_set_link_spring.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.tdsref_flatten.argtypes = [C.c_void_p, C.POINTER(tds_amd.Model)]
        L.tdsref_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.tdsref_debug.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.tdsref_hipstepper_selftest.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        if hasattr(L, "tdsref_hipstepper_selftest_env"):
            L.tdsref_hipstepper_selftest_env.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        if hasattr(L, "tdsref_generated_step"):
            L.tdsref_generated_step.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        if hasattr(L, "tdsref_rb_step"):
            L.tdsref_rb_step.argtypes = [C.POINTER(tds_amd.RbModel), C.c_int, C.c_int, C.c_void_p]
        if hasattr(L, "tdsref_rollout"):
            L.tdsref_rollout.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "tdsref_rollout_ex"):
            L.tdsref_rollout_ex.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 8
        if hasattr(L, "tdsref_hipstepper_selftest_devices"):
            L.tdsref_hipstepper_selftest_devices.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_void_p,
                                                             C.c_char_p, C.c_int]
        if hasattr(L, "tdsref_vecenv_steps"):
            L.tdsref_vecenv_steps.argtypes = [C.c_char_p, C.c_int, C.c_int] + [C.c_void_p] * 6
        if hasattr(L, "tdsref_vecenv_hip_worker"):
            L.tdsref_vecenv_hip_worker.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int] + \
                [C.c_void_p] * 5 + [C.c_char_p, C.c_int]
            L.tdsref_vecenv_hip_bench.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_int]
        if hasattr(L, "tdsref_vecenv_hip_lockstep"):
            L.tdsref_vecenv_hip_lockstep.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 3 + \
                [C.c_char_p, C.c_int]
        if hasattr(L, "tdsref_f32_create"):
            L.tdsref_f32_create.restype = C.c_void_p
            L.tdsref_f32_create.argtypes = [C.c_char_p, C.c_char_p, C.c_double]
            L.tdsref_f32_destroy.argtypes = [C.c_void_p]
            L.tdsref_f32_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class RefSimF32:
    """The reference's own FLOAT instantiation (TinyAlgebra<float, FloatUtils>, oracle/ref_harness_f32.cpp) of the
    step: name "ant" | "laikago" | "<file>.urdf[+plane]".  Host doubles in / out (rounded to float on entry)."""

    def __init__(self, name, in_dim, out_dim, dt=1e-3):
        self.h = C.c_void_p(lib().tdsref_f32_create(name.encode(), REF_ROOT.encode(), float(dt)))
        assert self.h, name
        self.in_dim, self.out_dim = int(in_dim), int(out_dim)

    def step(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.in_dim)
        y = np.zeros((x.shape[0], self.out_dim))
        lib().tdsref_f32_step(self.h, x.shape[0], self.in_dim, self.out_dim, x.ctypes.data, y.ctypes.data)
        return y

    def close(self):
        if self.h:
            lib().tdsref_f32_destroy(self.h)
            self.h = None


def hipstepper_selftest(batch=8, steps=5, env="ant"):
    """Reference VectorizedEnvironment<env> + tds_hip::HipStepper (include/tds_hip_stepper.hpp); env: "ant" |
    "laikago" | "humanoid".  returns (rc, message, obs0)."""
    obs0 = np.zeros(64)
    msg = C.create_string_buffer(512)
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        rc = lib().tdsref_hipstepper_selftest_env(env.encode(), batch, steps, obs0.ctypes.data, msg, 512)
        C.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    return rc, msg.value.decode(), obs0


def hipstepper_selftest_devices(devices, batch=10, steps=4):
    """HipStepper constructed with a device list (the batch split over one handle per entry) inside the reference's
    VectorizedEnvironment<Ant>; returns (rc, message)."""
    obs0 = np.zeros(64)
    msg = C.create_string_buffer(512)
    devs = (C.c_int * len(devices))(*devices)
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        rc = lib().tdsref_hipstepper_selftest_devices(batch, steps, len(devices), devs, obs0.ctypes.data, msg, 512)
        C.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    return rc, msg.value.decode()


def rollout(name, x0, params, steps, shift=0.0, network=None):
    """The reference's own rollout loop (Worker::rollouts + VectorizedEnvironment::policy/step, serial
    stepper) from the states x0 [B, dof_q+dof_qd] with per-environment linear policies params [B, P] — or, with
    network = (layer_sizes, activations, use_bias), the reference's NeuralNetwork with those layers per environment.
    returns (total_rewards [B], vec_steps [B] int32, final_obs [B, obs_dim])."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    b, od = x0.shape
    tot = np.zeros(b)
    cnt = np.zeros(b, dtype=np.int32)
    fin = np.zeros((b, od))
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        if network is not None:
            units, acts, bias = network
            assert len(acts) == len(units) - 1 and len(bias) == len(units)
            nn = np.array([len(units)] + list(units) + list(acts) + [1 if v else 0 for v in bias], dtype=np.int32)
            lib().tdsref_rollout_nn.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 6
            rc = lib().tdsref_rollout_nn(name.encode(), b, int(steps), float(shift), x0.ctypes.data, params.ctypes.data,
                                         nn.ctypes.data, tot.ctypes.data, cnt.ctypes.data, fin.ctypes.data)
        else:
            rc = lib().tdsref_rollout(name.encode(), b, int(steps), float(shift), x0.ctypes.data, params.ctypes.data,
                                      tot.ctypes.data, cnt.ctypes.data, fin.ctypes.data)
        C.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    if rc != 0:
        raise RuntimeError(f"tdsref_rollout({name}) failed: {rc}")
    return tot, cnt, fin


def generated_step(name, x, out_dim):
    """the reference's committed generated kernel omp_model_<name>_forward_zero_kernel (stateless)"""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros((x.shape[0], out_dim))
    rc = lib().tdsref_generated_step(name.encode(), x.shape[0], x.ctypes.data, y.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"no generated kernel for {name}")
    return y


def rb_step(model, state, steps=1):
    """the reference's World::step on RigidBody objects; state [n, num_bodies, 13]"""
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, model.num_bodies, 13)
    rc = lib().tdsref_rb_step(C.byref(model), st.shape[0], int(steps), st.ctypes.data)
    if rc:
        raise RuntimeError(f"tdsref_rb_step rc={rc}")
    return st


class RefSim:
    """name: "ant" | "laikago" | "<file>.urdf" | "<file>.urdf+plane" (file under <ref>/data)."""

    def __init__(self, name: str):
        # the reference printf()s "Loading URDF ..." to C stdout: silence it (fd-level) so that
        # callers that print machine-readable output (bench.py) stay clean
        L = lib()
        libc = C.CDLL(None)
        sys.stdout.flush()
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 1)
        try:
            self.h = L.tdsref_create(name.encode(), REF_ROOT.encode())
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        self.name = name
        self.input_dim = lib().tdsref_input_dim(self.h)
        self.output_dim = lib().tdsref_output_dim(self.h)

    def close(self):
        if self.h:
            lib().tdsref_destroy(self.h)
            self.h = None

    def last_penetrating_contacts(self):
        """per body pair of the reference's contact list, contacts with distance < 0 in the last step"""
        buf = (C.c_int * 16)()
        lib().tdsref_last_penetrating_contacts.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        n = lib().tdsref_last_penetrating_contacts(self.h, buf, 16)
        return [buf[i] for i in range(n)]

    def set_dt(self, dt):
        lib().tdsref_set_dt(self.h, dt)

    def set_gravity(self, g):
        lib().tdsref_set_gravity(self.h, *map(float, g))

    def set_solver(self, cfm, erp, pgs_iterations=1, friction=1.0, restitution=0.0):
        lib().tdsref_set_solver(self.h, cfm, erp, pgs_iterations, friction, restitution)

    def set_link_spring(self, link, stiffness, damping):
        if lib().tdsref_set_link_spring(self.h, link, stiffness, damping) != 0:
            raise RuntimeError("tdsref_set_link_spring: no such link")

    def flatten(self) -> "tds_amd.Model":
        m = tds_amd.Model()
        rc = lib().tdsref_flatten(self.h, C.byref(m))
        if rc != 0:
            raise RuntimeError(f"tdsref_flatten failed rc={rc}")
        return m

    def step(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.input_dim)
        y = np.zeros((x.shape[0], self.output_dim), dtype=np.float64)
        lib().tdsref_step(self.h, x.shape[0], x.ctypes.data, y.ctypes.data)
        return y

    def debug(self, x: np.ndarray, model):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        nd, nl = model.dof_qd, model.num_links
        ncmax = 64
        qdd = np.zeros(nd)
        M = np.zeros((nd, nd))
        contacts = np.zeros((ncmax, 10))
        jac = np.zeros((ncmax, 3, nd))
        links = np.zeros(ncmax, dtype=np.int32)
        Xw = np.zeros((nl, 12))
        nc = lib().tdsref_debug(self.h, x.ctypes.data, qdd.ctypes.data, M.ctypes.data,
                                contacts.ctypes.data, jac.ctypes.data, links.ctypes.data,
                                Xw.ctypes.data)
        return dict(qdd=qdd, M=M, contacts=contacts[:nc], jac=jac[:nc], links=links[:nc],
                    X_world=Xw)


def vecenv_steps(name, x0, actions, output_dim):
    """The reference's VectorizedEnvironment::step driven like its Python binding (auto_reset_when_done = false,
    serial stepper) from the states x0 [B, obs_dim] with the action sequence actions [T, B, action_dim].
    returns obs [T, B, obs_dim], rewards [T, B], dones [T, B], visual_world_transforms [T, B, output_dim]."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    actions = np.ascontiguousarray(actions, dtype=np.float64)
    T, B, _ = actions.shape
    od = x0.shape[1]
    obs = np.zeros((T, B, od))
    rew = np.zeros((T, B))
    done = np.zeros((T, B))
    vis = np.zeros((T, B, output_dim))
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        rc = lib().tdsref_vecenv_steps(name.encode(), B, T, x0.ctypes.data, actions.ctypes.data, obs.ctypes.data,
                                       rew.ctypes.data, done.ctypes.data, vis.ctypes.data)
        C.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    assert rc == 0, rc
    return obs, rew, done, vis


def rollout_ex(name, x0, params, steps, shift, output_dim):
    """reflib.rollout plus the by-products of Worker::rollouts: RunningStat (count, mean, S) per environment and
    observation component [B, obs_dim, 3], trajectories [B, steps, output_dim] and their lengths [B]."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    params = np.ascontiguousarray(params, dtype=np.float64)
    b, od = x0.shape
    tot = np.zeros(b)
    cnt = np.zeros(b, dtype=np.int32)
    fin = np.zeros((b, od))
    stats = np.zeros((b, od, 3))
    traj = np.zeros((b, steps, output_dim))
    tlen = np.zeros(b, dtype=np.int32)
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        rc = lib().tdsref_rollout_ex(name.encode(), b, steps, float(shift), x0.ctypes.data, params.ctypes.data,
                                     tot.ctypes.data, cnt.ctypes.data, fin.ctypes.data, stats.ctypes.data,
                                     traj.ctypes.data, tlen.ctypes.data)
        C.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    assert rc == 0, rc
    return tot, cnt, fin, stats, traj, tlen


def vecenv_hip_worker(name, batch, steps, params, output_dim, shift=0.0, seed=12345, auto_reset=False):
    """The reference's Worker<Env>::rollouts (ars_vectorized_worker.h:51-140, unmodified header) instantiated on the
    reference's VectorizedEnvironment (CPU, serial stepper) AND on tds_hip::VectorizedEnv (environments resident on
    the GPU, include/tds_hip_stepper.hpp), same seed / policies.  Returns a dict of [2, batch, ...] arrays: index 0
    the reference, index 1 the HIP class."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    tr = np.zeros((2, batch))
    vs = np.zeros((2, batch), dtype=np.int32)
    last = np.zeros((2, batch, output_dim))
    tl = np.zeros((2, batch), dtype=np.int32)
    msg = C.create_string_buffer(512)
    rc = lib().tdsref_vecenv_hip_worker(name.encode(), int(batch), int(steps), float(shift), int(seed), int(bool(auto_reset)),
                                        params.ctypes.data, tr.ctypes.data, vs.ctypes.data, last.ctypes.data, tl.ctypes.data,
                                        msg, 512)
    if rc != 0:
        raise RuntimeError(f"tdsref_vecenv_hip_worker: {rc} {msg.value.decode()}")
    return {"total_rewards": tr, "vec_steps": vs, "traj_last": last, "traj_len": tl}


def vecenv_hip_lockstep(name, batch, steps, params, seed=12345, auto_reset=False):
    """The reference's VectorizedEnvironment and tds_hip::VectorizedEnv stepped in lock step under the environments' own
    linear policies, the HIP class's state re-synchronised to the reference's after every step: per environment the
    largest |hip - ref| / max(1, |ref|) over every observation, reward, y record and state of every step."""
    params = np.ascontiguousarray(params, dtype=np.float64)
    worst = np.zeros(batch)
    counts = np.zeros(3, dtype=np.int64)
    msg = C.create_string_buffer(512)
    rc = lib().tdsref_vecenv_hip_lockstep(name.encode(), int(batch), int(steps), int(seed), int(bool(auto_reset)),
                                          params.ctypes.data, worst.ctypes.data, counts.ctypes.data, msg, 512)
    if rc != 0:
        raise RuntimeError(f"tdsref_vecenv_hip_lockstep: {rc} {msg.value.decode()}")
    return {"worst": worst, "done_mismatches": int(counts[0]), "host_resets": int(counts[1]), "values_compared": int(counts[2])}


def vecenv_hip_bench(name, batch, steps):
    """env-steps/s of tds_hip::VectorizedEnv driven from C++: step() with / without the y records, step_many_device
    (per-step records into device rings), rollouts_on_device"""
    rates = np.zeros(4)
    msg = C.create_string_buffer(512)
    rc = lib().tdsref_vecenv_hip_bench(name.encode(), int(batch), int(steps), rates.ctypes.data, msg, 512)
    if rc != 0:
        raise RuntimeError(f"tdsref_vecenv_hip_bench: {rc} {msg.value.decode()}")
    return {"step_host_vectors": rates[0], "step_host_vectors_no_graphics": rates[1], "step_many_device": rates[2],
            "rollouts_on_device": rates[3]}
