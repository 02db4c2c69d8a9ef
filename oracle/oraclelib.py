"""TEST INFRASTRUCTURE.  ctypes access to oracle/libtds_oracle.so (the plain-C restatement,
tds_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import tds_amd  # noqa: E402

_LIB_PATH = os.path.join(_HERE, "libtds_oracle.so")


class _Debug(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("qdd", "M", "Minv", "contacts", "jac", "lcp_A", "lcp_b", "lcp_p", "X_world")] + \
               [("n_c", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libtds_oracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER(tds_amd.Model)
        L.tds_oracle_step.argtypes = [P, C.c_int, C.c_void_p, C.c_void_p]
        L.tds_oracle_step_omp.argtypes = [P, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.tds_oracle_step_debug.argtypes = [P, C.c_void_p, C.c_void_p, C.POINTER(_Debug)]
        L.tds_oracle_rb_step.argtypes = [C.POINTER(tds_amd.RbModel), C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def step(model, x, threads=1):
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, model.input_dim)
    y = np.zeros((x.shape[0], model.output_dim))
    if threads == 1:
        rc = lib().tds_oracle_step(C.byref(model), x.shape[0], x.ctypes.data, y.ctypes.data)
    else:
        rc = lib().tds_oracle_step_omp(C.byref(model), x.shape[0], x.ctypes.data, y.ctypes.data,
                                       threads)
    if rc:
        raise RuntimeError(f"tds_oracle_step rc={rc}")
    return y


def rb_step(model, state, steps=1):
    """free rigid bodies (row a20): state [n, num_bodies, 13] advanced by `steps` World::step calls"""
    st = np.array(state, dtype=np.float64, order="C", copy=True).reshape(-1, model.num_bodies, 13)
    rc = lib().tds_oracle_rb_step(C.byref(model), st.shape[0], int(steps), st.ctypes.data)
    if rc:
        raise RuntimeError(f"tds_oracle_rb_step rc={rc}")
    return st


def max_threads():
    return lib().tds_oracle_max_threads()


def step_debug(model, x):
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    nd, nl, ncm = model.dof_qd, model.num_links, 64
    out = dict(qdd=np.zeros(nd), M=np.zeros((nd, nd)), Minv=np.zeros((nd, nd)),
               contacts=np.zeros((ncm, 10)), jac=np.zeros((ncm, 3, nd)),
               lcp_A=np.zeros(9 * ncm * ncm), lcp_b=np.zeros(3 * ncm), lcp_p=np.zeros(3 * ncm),
               X_world=np.zeros((nl, 12)))
    d = _Debug()
    for k, v in out.items():
        setattr(d, k, v.ctypes.data)
    y = np.zeros(model.output_dim)
    rc = lib().tds_oracle_step_debug(C.byref(model), x.ctypes.data, y.ctypes.data, C.byref(d))
    if rc:
        raise RuntimeError(f"tds_oracle_step_debug rc={rc}")
    nc = d.n_c
    out["contacts"] = out["contacts"][:nc]
    out["jac"] = out["jac"][:nc]
    out["lcp_A"] = out["lcp_A"][:9 * nc * nc].reshape(3 * nc, 3 * nc)
    out["lcp_b"] = out["lcp_b"][:3 * nc]
    out["lcp_p"] = out["lcp_p"][:3 * nc]
    out["y"] = y
    return out
