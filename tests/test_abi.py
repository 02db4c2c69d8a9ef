"""CPU: the C-ABI library loads and exports every symbol include/tds_hip.h declares; host-side
model handling; loud failure without a GPU (no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import pytest

from conftest import MODELS, ROOT

import tds_amd
from tds_amd import hip_backend


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "tds_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tds_(?:hip|rb)_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(built):
    L = hip_backend.lib()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(hip_backend.EXPORTED_SYMBOLS) == syms
    assert L.tds_hip_abi_version() == tds_amd.TDS_HIP_ABI_VERSION


def test_struct_layout_matches_c(built):
    """sizeof(tds_model_t) as seen by C must equal the ctypes mirror."""
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include "tds_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu",sizeof(tds_model_t),sizeof(tds_link_t),sizeof(tds_geom_t),sizeof(tds_visual_t),sizeof(tds_rb_model_t),sizeof(tds_rb_body_t),sizeof(tds_body_t));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    assert [int(v) for v in out] == [C.sizeof(tds_amd.Model), C.sizeof(tds_amd.model.Link),
                                     C.sizeof(tds_amd.model.Geom), C.sizeof(tds_amd.model.Visual),
                                     C.sizeof(tds_amd.RbModel), C.sizeof(tds_amd.RbBody), C.sizeof(tds_amd.model.Body)]


@pytest.mark.parametrize("name", MODELS)
def test_models_pass_model_check(name, built):
    m = tds_amd.load_model(name)
    hip_backend.model_check(m)
    assert m.input_dim > 0 and m.output_dim > 0
    # JSON round trip
    assert tds_amd.model_to_dict(tds_amd.model_from_dict(tds_amd.model_to_dict(m))) == tds_amd.model_to_dict(m)


def test_model_check_rejects_unsupported(built):
    m = tds_amd.load_model("ant")
    m.is_floating = 1
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)
    m = tds_amd.load_model("ant")
    m.links[7].joint_type = tds_amd.model.JOINT_SPHERICAL
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)
    m = tds_amd.load_model("ant")
    m.abi_version = 99
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)


def test_no_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = tds_amd.load_model("ant")
    h = C.c_void_p()
    rc = hip_backend.lib().tds_hip_create(C.byref(m), 8, 0, 0, C.byref(h))
    assert rc in (3, 4) and not h.value  # TDS_ERR_HIP / TDS_ERR_NO_DEVICE, never a silent CPU path
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.HipSim(m, 8)


@pytest.mark.parametrize("name", ["ant", "laikago", "humanoid"])
def test_legacy_shim_exports_reference_symbols(name, built):
    """cuda_model_<env>.so must export exactly what CudaModel<double> dlsym()s
    (reference: examples/ars/ars_train_policy_cuda.cpp:220-229, 345-359)."""
    import torch  # noqa: F401  one HIP runtime per process
    path = os.path.join(ROOT, "tiny-differentiable-simulator_amd", f"cuda_model_{name}.so")
    L = C.CDLL(path)

    class Meta(C.Structure):
        _fields_ = [("output_dim", C.c_int), ("input_dim", C.c_int), ("global_dim", C.c_int)]

    base = f"cuda_model_{name}_forward_zero"
    for suffix in ("", "_meta", "_allocate", "_deallocate"):
        assert hasattr(L, base + suffix)
    meta = getattr(L, base + "_meta")
    meta.restype = Meta
    md = meta()
    m = tds_amd.load_model(name)
    assert (md.output_dim, md.input_dim, md.global_dim) == (m.output_dim, m.input_dim, 0)


@pytest.mark.parametrize("name", ["ant", "laikago", "humanoid"])
def test_newer_abi_library_exports_reference_symbols(name, built):
    """cudalib_<env>.so must export what CudaLibrary<double> / CudaFunction<double> dlsym()
    (reference: src/utils/cuda/cuda_library.hpp:50-56, cuda_function.hpp:78-99)."""
    import torch  # noqa: F401  one HIP runtime per process
    L = C.CDLL(os.path.join(ROOT, "tiny-differentiable-simulator_amd", f"cudalib_{name}.so"))

    class Meta(C.Structure):
        _fields_ = [("output_dim", C.c_int), ("local_input_dim", C.c_int),
                    ("global_input_dim", C.c_int), ("accumulated_output", C.c_bool)]

    names = C.POINTER(C.c_char_p)()
    count = C.c_int(0)
    L.model_info(C.byref(names), C.byref(count))
    assert count.value == 1 and names[0].decode() == f"cuda_model_{name}"
    base = names[0].decode() + "_forward_zero"
    for suffix in ("", "_meta", "_allocate", "_deallocate", "_send_local", "_send_global"):
        assert hasattr(L, base + suffix)
    meta = getattr(L, base + "_meta")
    meta.restype = Meta
    md = meta()
    m = tds_amd.load_model(name)
    assert (md.output_dim, md.local_input_dim, md.global_input_dim, md.accumulated_output) == \
        (m.output_dim, m.input_dim, 0, False)


def test_model_check_on_multi_body_worlds(built):
    """CPU: the multi-body blobs pass (fixed and floating bases; kernels of kind 3 / 4), malformed ones are refused with
    a reason"""
    for name in ["three_pendulums_plane", "four_pendulums", "two_cubes_floating", "pendulum_and_cube"]:
        hip_backend.model_check(tds_amd.load_model(name))
    m = tds_amd.load_model("three_pendulums")
    m.bodies[2].first_link = m.bodies[1].first_link - 1          # not ascending
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)
    m = tds_amd.load_model("three_pendulums")
    m.links[7].parent = 2                                        # a link of body 1 hanging off a link of body 0
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)
    m = tds_amd.load_model("three_pendulums")
    m.num_bodies = 5                                             # more than TDS_MAX_BODIES
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)
    m = tds_amd.load_model("pendulum_and_cube")
    m.links[2].joint_type = tds_amd.model.JOINT_SPHERICAL        # spherical joints are not taken in multi-body worlds
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)
    m = tds_amd.load_model("two_cubes_floating")
    m.dof_q -= 1                                                 # dof_q inconsistent with two floating bases
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)
    m = tds_amd.load_model("two_pendulums")
    m.geoms[0].link = 7                                          # a geometry of body 0 listed on a link of body 1
    with pytest.raises(hip_backend.TdsHipError):
        hip_backend.model_check(m)
