"""ctypes binding of libtds_hip.so (C ABI: include/tds_hip.h).

PyTorch is used for what it is good at here — device memory, streams, torch.distributed —
never for the arithmetic: every step goes through the hand-written HIP kernel.  If the
shared library is missing or no HIP device is visible the constructors raise; there is no
CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

from . import model as _model

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TDS_HIP_LIB", os.path.join(_HERE, "libtds_hip.so"))

TDS_OK = 0

_lib = None


class TdsHipError(RuntimeError):
    pass


def lib():
    """Load libtds_hip.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TdsHipError(f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  Import torch
        # first so that libtds_hip.so binds to the runtime torch already loaded (same streams,
        # same device contexts) instead of pulling a second copy from /opt/rocm.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        P = C.POINTER(_model.Model)
        L.tds_hip_last_error.restype = C.c_char_p
        L.tds_hip_model_check.argtypes = [P]
        L.tds_hip_create.argtypes = [P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.tds_hip_destroy.argtypes = [C.c_void_p]
        L.tds_hip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        for f in ("tds_hip_num_envs", "tds_hip_input_dim", "tds_hip_output_dim", "tds_hip_dtype"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.tds_hip_x_device.restype = C.c_void_p
        L.tds_hip_x_device.argtypes = [C.c_void_p]
        L.tds_hip_y_device.restype = C.c_void_p
        L.tds_hip_y_device.argtypes = [C.c_void_p]
        L.tds_hip_set_inputs.argtypes = [C.c_void_p, C.c_void_p]
        L.tds_hip_get_inputs.argtypes = [C.c_void_p, C.c_void_p]
        L.tds_hip_get_outputs.argtypes = [C.c_void_p, C.c_void_p]
        L.tds_hip_forward_zero_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.tds_hip_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.tds_hip_step_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.tds_hip_obs_dim.argtypes = [C.c_void_p]
        L.tds_hip_set_auto_reset.argtypes = [C.c_void_p, C.c_int, C.c_ulonglong]
        L.tds_hip_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.tds_hip_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.tds_hip_forward_zero_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.tds_hip_rollout_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int] + [C.c_void_p] * 6
        L.tds_rb_last_error.restype = C.c_char_p
        L.tds_rb_create.argtypes = [C.POINTER(_model.RbModel), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.tds_rb_destroy.argtypes = [C.c_void_p]
        L.tds_rb_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.tds_rb_state_device.argtypes = [C.c_void_p]
        L.tds_rb_state_device.restype = C.c_void_p
        L.tds_rb_set_state.argtypes = [C.c_void_p, C.c_void_p]
        L.tds_rb_get_state.argtypes = [C.c_void_p, C.c_void_p]
        L.tds_rb_step.argtypes = [C.c_void_p, C.c_int]
        L.tds_hip_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.tds_hip_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.tds_hip_profile_phases.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
        L.tds_hip_kernel_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for f in ("tds_hip_device", "tds_hip_record_bytes", "tds_hip_sync", "tds_hip_forward_zero_host_end"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.tds_hip_forward_zero_host_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        for f in ("tds_hip_step_many_prepare", "tds_hip_step_many"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.tds_hip_set_graph_chains.argtypes = [C.c_void_p, C.c_int]
        for f in ("tds_hip_step_many_rings", "tds_hip_step_many_rings_prepare"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Rings)]
        L.tds_hip_step_many_rings_blocks.argtypes = [C.c_void_p]
        L.tds_hip_step_many_is_loop.argtypes = [C.c_void_p, C.c_int]
        L.tds_hip_debug_poison_lds.argtypes = [C.c_void_p, C.c_int]
        L.tds_hip_step_many_tune.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
        # multi-GPU shards (RCCL all-gather of the observation records)
        L.tds_hip_shard_unique_id.argtypes = [C.c_void_p]
        L.tds_hip_shard_create.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.POINTER(C.c_void_p)]
        L.tds_hip_shard_create_all.argtypes = [P, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                                               C.POINTER(C.c_void_p)]
        L.tds_hip_shard_sim.restype = C.c_void_p
        for f in ("tds_hip_shard_destroy", "tds_hip_shard_sim", "tds_hip_shard_rank", "tds_hip_shard_world",
                  "tds_hip_shard_local_envs", "tds_hip_shard_first_env", "tds_hip_shard_wire_bytes",
                  "tds_hip_shard_flush"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.tds_hip_shard_set_block.argtypes = [C.c_void_p, C.c_int]
        L.tds_hip_shard_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.tds_hip_shard_step_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.tds_hip_shard_step_many_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.tds_hip_shard_group_step.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int]
        L.tds_hip_shard_gathered.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
        L.tds_hip_shard_ring_plan.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        L.tds_hip_shard_gathered_offset.argtypes = [C.c_int, C.c_int, C.c_int]
        L.tds_hip_shard_gathered_offset.restype = C.c_longlong
        L.tds_hip_default_option.argtypes = [C.c_char_p, C.c_longlong]
        L.tds_hip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
        L.tds_hip_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
        L.tds_hip_option_name.argtypes = [C.c_int]
        L.tds_hip_option_name.restype = C.c_char_p
        _lib = L
    return _lib


class Rings(C.Structure):
    """tds_hip_rings_t (include/tds_hip.h): per-step record rings of tds_hip_step_many_rings"""
    _fields_ = [("obs_ring", C.c_void_p), ("obs_slots", C.c_int32), ("obs_first", C.c_int32), ("obs_f32", C.c_int32),
                ("y_stride", C.c_int32), ("y_ring", C.c_void_p), ("y_slots", C.c_int32), ("y_first", C.c_int32),
                ("progress", C.c_void_p), ("obs_slot_envs", C.c_int32), ("pad1_", C.c_int32)]


EXPORTED_SYMBOLS = [
    "tds_hip_last_error", "tds_hip_abi_version", "tds_hip_device_count", "tds_hip_model_check",
    "tds_hip_create", "tds_hip_destroy", "tds_hip_set_stream", "tds_hip_num_envs",
    "tds_hip_input_dim", "tds_hip_output_dim", "tds_hip_dtype", "tds_hip_x_device",
    "tds_hip_y_device", "tds_hip_set_inputs", "tds_hip_get_inputs", "tds_hip_get_outputs",
    "tds_hip_forward_zero_device", "tds_hip_step", "tds_hip_step_obs", "tds_hip_obs_dim",
    "tds_hip_set_auto_reset", "tds_hip_reset", "tds_hip_rollout", "tds_hip_rollout_ex",
    "tds_hip_set_policy_network", "tds_hip_policy_num_parameters",
    "tds_hip_forward_zero_host", "tds_hip_send_local", "tds_hip_forward_zero_fetch",
    "tds_hip_set_timing", "tds_hip_last_kernel_ms", "tds_hip_kernel_info", "tds_hip_profile_phases",
    "tds_hip_device", "tds_hip_record_bytes", "tds_hip_sync", "tds_hip_forward_zero_host_begin",
    "tds_hip_forward_zero_host_end", "tds_hip_step_many_prepare", "tds_hip_step_many",
    "tds_hip_set_graph_chains", "tds_hip_step_many_tune", "tds_hip_step_many_is_loop", "tds_hip_debug_poison_lds",
    "tds_hip_step_many_rings", "tds_hip_step_many_rings_prepare", "tds_hip_step_many_rings_blocks",
    "tds_hip_default_option", "tds_hip_set_option", "tds_hip_get_option", "tds_hip_option_count", "tds_hip_option_name",
    "tds_hip_profile_zones", "tds_hip_step_host", "tds_hip_reset_host", "tds_hip_set_states", "tds_hip_device_alloc", "tds_hip_device_free",
    "tds_hip_device_upload", "tds_hip_device_download",
    "tds_hip_shard_rccl_version", "tds_hip_shard_unique_id", "tds_hip_shard_create", "tds_hip_shard_create_all",
    "tds_hip_shard_destroy", "tds_hip_shard_sim", "tds_hip_shard_rank", "tds_hip_shard_world",
    "tds_hip_shard_local_envs", "tds_hip_shard_first_env", "tds_hip_shard_wire_bytes", "tds_hip_shard_set_block",
    "tds_hip_shard_step", "tds_hip_shard_step_many", "tds_hip_shard_step_many_prepare", "tds_hip_shard_group_step", "tds_hip_shard_flush", "tds_hip_shard_gathered", "tds_hip_shard_gathered_step",
    "tds_hip_shard_ring_plan", "tds_hip_shard_gathered_offset", "tds_hip_shard_exchange_form", "tds_hip_shard_peer_count",
    "tds_hip_single_step_kernel",
    "tds_rb_last_error", "tds_rb_create", "tds_rb_destroy", "tds_rb_set_stream", "tds_rb_state_device",
    "tds_rb_set_state", "tds_rb_get_state", "tds_rb_step",
]


def lib_has_symbol(name: str) -> bool:
    """whether libtds_hip.so exports `name` (e.g. the entry point of an experiment slot, tools/build_alt.sh)"""
    return hasattr(lib(), name)


def shard_ring_plan(chunks_done: int, n_steps: int, act_first: int = 0, act_blocks: int = 1, n_blocks: int = 1):
    """the step-loop launches a tds_hip_shard_step_many call is cut into (tds_hip_shard_ring_plan; no device needed):
    list of dicts half / steps / step0 / act_first / slot0 / first_wait"""
    cap = 80
    out = (C.c_int * (6 * cap))()
    n = lib().tds_hip_shard_ring_plan(int(chunks_done), int(n_steps), int(act_first), int(act_blocks), int(n_blocks), out, cap)
    if n < 0:
        raise TdsHipError("tds_hip_shard_ring_plan: bad arguments")
    keys = ("half", "steps", "step0", "act_first", "slot0", "first_wait")
    return [dict(zip(keys, out[6 * i:6 * i + 6])) for i in range(n)]


def default_option(key: str, value) -> None:
    """process-wide default of a library option for handles created from now on (tds_hip_default_option;
    None: back to "unset" = environment variable TDS_HIP_<KEY>, else the library's own rule)"""
    v = -(1 << 63) if value is None else int(value)
    _check(lib().tds_hip_default_option(key.encode(), v))


def option_names():
    L = lib()
    return [L.tds_hip_option_name(i).decode() for i in range(L.tds_hip_option_count())]


class default_options:
    """``with default_options(w2=0, no_chain=1): sim = HipSim(...)`` — create-time options for the handles made inside"""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        for k, v in self.kv.items():
            default_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kv:
            default_option(k, None)
        return False


def _check(rc):
    if rc != TDS_OK:
        raise TdsHipError(f"tds_hip error {rc}: {lib().tds_hip_last_error().decode()}")


def model_check(m: _model.Model) -> None:
    _check(lib().tds_hip_model_check(C.byref(m)))


def wrap_device_pointer(ptr: int, shape, torch_dtype, device: int, owner=None):
    """zero-copy torch view of library-owned device memory"""
    import torch

    class _Holder:
        pass

    hld = _Holder()
    hld.__cuda_array_interface__ = {
        "shape": tuple(shape), "typestr": "<f8" if torch_dtype == torch.float64 else "<f4",
        "data": (int(ptr), False), "version": 2, "strides": None,
    }
    t = torch.as_tensor(hld, device=f"cuda:{device}")
    t._tds_owner = owner
    return t


def dtype_code(dtype) -> int:
    """"f64": double arithmetic + double records; "mixed": double arithmetic + FLOAT records (the reference's float
    record ABI, parity-gated); "f32": pure float (measured only)."""
    if isinstance(dtype, int):
        return dtype
    return {"f64": _model.TDS_DTYPE_F64, "float64": _model.TDS_DTYPE_F64, "f32": _model.TDS_DTYPE_F32,
            "float32": _model.TDS_DTYPE_F32, "mixed": _model.TDS_DTYPE_F64_REC32,
            "f64r32": _model.TDS_DTYPE_F64_REC32}[dtype]


class HipSim:
    """N resident environments of one model on one GPU.

    ``x`` / ``y`` are torch views (no copy) of the library-owned env-major records
    [N, input_dim] / [N, output_dim] in the RECORD dtype (``torch_dtype``: float64 for "f64", float32 for "mixed"
    and "f32").
    """

    def __init__(self, m: _model.Model, num_envs: int, device: int = 0, dtype: str = "f64",
                 lanes_per_env: int | None = None, na_cap: int | None = None, _handle=None, _owner=None,
                 options: dict | None = None):
        import torch

        if not torch.cuda.is_available():
            raise TdsHipError("no HIP device visible (the HIP path has no CPU fallback)")
        self.model = m.copy()
        self.num_envs = int(num_envs)
        self.device = int(device)
        self.dtype = dtype_code(dtype)
        self.torch_dtype = torch.float64 if self.dtype == _model.TDS_DTYPE_F64 else torch.float32
        self._owner = _owner  # a HipShard owns the handle of its sim
        if _handle is not None:
            self.h = _handle
            self.input_dim = self.model.input_dim
            self.output_dim = self.model.output_dim
            self.x = self._wrap(lib().tds_hip_x_device(self.h), (self.num_envs, self.input_dim))
            self.y = self._wrap(lib().tds_hip_y_device(self.h), (self.num_envs, self.output_dim))
            self.use_current_stream()
            return
        create_opts = {}
        if lanes_per_env is not None:
            create_opts["lanes_per_env"] = lanes_per_env
        if na_cap is not None:
            create_opts["na_cap"] = na_cap
        if options:  # create-time options go through the process defaults, run-time ones are set on the new handle
            ct = ("lanes_per_env", "na_cap", "w2", "gram", "no_chain", "no_rootjoint", "no_kinchain", "no_eulerroot",
                  "no_legscan", "fold_fixed", "quad", "oct", "chain")
            create_opts.update({k: v for k, v in options.items() if k in ct})
        h = C.c_void_p()
        with default_options(**create_opts):
            _check(lib().tds_hip_create(C.byref(self.model), self.num_envs, self.device, self.dtype, C.byref(h)))
        self.h = h
        if options:
            for k, v in options.items():
                if k not in create_opts:
                    self.set_option(k, v)
        self.input_dim = self.model.input_dim
        self.output_dim = self.model.output_dim
        self.x = self._wrap(lib().tds_hip_x_device(self.h), (self.num_envs, self.input_dim))
        self.y = self._wrap(lib().tds_hip_y_device(self.h), (self.num_envs, self.output_dim))
        self.use_current_stream()

    # -- zero-copy torch view of library-owned device memory -------------------------------
    def _wrap(self, ptr, shape):
        import torch

        n = 1
        for s in shape:
            n *= s
        itemsize = 8 if getattr(self, "dtype", _model.TDS_DTYPE_F64) == _model.TDS_DTYPE_F64 else 4
        typestr = "<f8" if itemsize == 8 else "<f4"

        class _Holder:
            pass

        hld = _Holder()
        hld.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2,
            "strides": None,
        }
        t = torch.as_tensor(hld, device=f"cuda:{self.device}")
        t._tds_owner = self  # keep the handle alive as long as the view lives
        return t

    def set_option(self, key: str, value) -> None:
        """run-time option of this handle (tds_hip_set_option; csrc/tds_options.h lists the keys)"""
        _check(lib().tds_hip_set_option(self.h, key.encode(), -(1 << 63) if value is None else int(value)))

    def get_option(self, key: str):
        """the option's value, or None while it is unset (library rule)"""
        v, st = C.c_longlong(0), C.c_int(0)
        _check(lib().tds_hip_get_option(self.h, key.encode(), C.byref(v), C.byref(st)))
        return int(v.value) if st.value else None

    def use_current_stream(self):
        import torch

        st = torch.cuda.current_stream(self.device)
        _check(lib().tds_hip_set_stream(self.h, C.c_void_p(st.cuda_stream)))

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_owner", None) is None:
                lib().tds_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step_many_prepare(self, actions, n_steps: int, obs=None, first_block: int = 0):
        """Capture + instantiate the hipGraph of ``n_steps`` closed-loop steps without running anything."""
        ap, nb, op = self._many_args(actions, obs)
        _check(lib().tds_hip_step_many_prepare(self.h, ap, nb, int(first_block), int(n_steps), op))

    def step_many(self, actions, n_steps: int, obs=None, first_block: int = 0):
        """``n_steps`` closed-loop steps per host call (tds_hip_step_many: chained hipGraphs or one step-loop launch,
        see step_many_is_loop): step k takes the action block ``actions[(first_block + k) % len(actions)]``
        ([B, N, action_dim] device tensor, or None).  With auto-reset on, every step resets what it ends with done."""
        ap, nb, op = self._many_args(actions, obs)
        _check(lib().tds_hip_step_many(self.h, ap, nb, int(first_block), int(n_steps), op))

    def _rings(self, obs_ring, y_ring, obs_first, y_first, progress=None):
        import torch

        r = Rings()
        if obs_ring is not None:
            assert obs_ring.is_cuda and obs_ring.is_contiguous() and obs_ring.dim() == 3
            assert tuple(obs_ring.shape[1:]) == (self.num_envs, self.obs_dim + 2)
            assert obs_ring.dtype in (self.torch_dtype, torch.float32)
            r.obs_ring, r.obs_slots, r.obs_first = obs_ring.data_ptr(), int(obs_ring.shape[0]), int(obs_first)
            r.obs_f32 = 1 if (obs_ring.dtype == torch.float32 and self.torch_dtype != torch.float32) else 0
        if y_ring is not None:
            assert y_ring.is_cuda and y_ring.is_contiguous() and y_ring.dim() == 3 and y_ring.dtype == self.torch_dtype
            # (a last dimension beyond output_dim: a padded record stride, tds_hip_rings_t::y_stride)
            assert int(y_ring.shape[1]) == self.num_envs and int(y_ring.shape[2]) >= self.output_dim
            r.y_ring, r.y_slots, r.y_first = y_ring.data_ptr(), int(y_ring.shape[0]), int(y_first)
            r.y_stride = int(y_ring.shape[2]) if int(y_ring.shape[2]) != self.output_dim else 0
        if progress is not None:
            # (one counter per slot of the obs ring, tds_hip_rings_t::progress)
            assert progress.is_cuda and progress.dtype == torch.int64 and obs_ring is not None
            assert progress.numel() >= int(obs_ring.shape[0])
            r.progress = progress.data_ptr()
        return r

    def step_many_rings(self, actions, n_steps: int, obs_ring=None, y_ring=None, first_block: int = 0,
                        obs_first: int = 0, y_first: int = 0, progress=None, prepare_only: bool = False):
        """``n_steps`` closed-loop steps per host call WITH per-step records (tds_hip_step_many_rings): step k leaves its
        [obs | reward | done] record in ``obs_ring[(obs_first + k) % len(obs_ring)]`` ([S, N, obs_dim + 2]) and its y
        record in ``y_ring[(y_first + k) % len(y_ring)]`` ([S', N, output_dim]) — what the reference's
        VectorizedEnvironment::step hands out every step.  One step-loop launch where step_many_is_loop holds."""
        ap, nb, _ = self._many_args(actions, None)
        r = self._rings(obs_ring, y_ring, obs_first, y_first, progress)
        f = lib().tds_hip_step_many_rings_prepare if prepare_only else lib().tds_hip_step_many_rings
        _check(f(self.h, ap, nb, int(first_block), int(n_steps), C.byref(r)))

    def prepared_step_many_rings(self, actions, n_steps: int, obs_ring=None, y_ring=None, first_block: int = 0,
                                 obs_first: int = 0, y_first: int = 0):
        """step_many_rings with every argument marshalled NOW: returns a callable whose body is the one C call (a
        20-step timed region is 0.3 ms — tens of microseconds of Python argument checking inside it are a tenth of it).
        Builds the graphs of the graph form as well (tds_hip_step_many_rings_prepare)."""
        ap, nb, _ = self._many_args(actions, None)
        r = self._rings(obs_ring, y_ring, obs_first, y_first, None)
        L, h, fb, ns, rr = lib(), self.h, int(first_block), int(n_steps), C.byref(r)
        _check(L.tds_hip_step_many_rings_prepare(h, ap, nb, fb, ns, rr))
        keep = (actions, obs_ring, y_ring, r)  # (the tensors and the struct must outlive the callable)

        def call(_f=L.tds_hip_step_many_rings, _keep=keep):
            if _f(h, ap, nb, fb, ns, rr) != TDS_OK:
                _check(-1)

        return call

    def step_many_rings_raw(self, actions, n_steps: int, rings: "Rings", first_block: int = 0):
        """tds_hip_step_many_rings with a hand-filled tds_hip_rings_t (obs_slot_envs, strides ...)"""
        ap, nb, _ = self._many_args(actions, None)
        _check(lib().tds_hip_step_many_rings(self.h, ap, nb, int(first_block), int(n_steps), C.byref(rings)))

    def profile_zones(self) -> dict:
        """one instrumented step, reported through the SubmitProfileTiming-shaped callback (tds_hip_profile_zones):
        {zone name: microseconds}"""
        out = {}
        FN = C.CFUNCTYPE(None, C.c_char_p, C.c_double, C.c_void_p)

        def cb(name, us, _user):
            out[name.decode()] = float(us)

        f = FN(cb)
        lib().tds_hip_profile_zones.argtypes = [C.c_void_p, FN, C.c_void_p]
        _check(lib().tds_hip_profile_zones(self.h, f, None))
        return out

    def rings_blocks(self) -> int:
        """increments of a rings progress counter per completed step (= workgroups of the step-loop launch)"""
        return int(lib().tds_hip_step_many_rings_blocks(self.h))

    def debug_poison_lds(self, byte_pattern: int = 0xFF):
        """Test aid: every compute unit's LDS filled with the byte pattern (0xFF: NaN in every scalar type)."""
        _check(lib().tds_hip_debug_poison_lds(self.h, int(byte_pattern)))

    def step_many_is_loop(self, n_steps: int) -> bool:
        """True if step_many(n_steps) runs as launches of the step-loop kernel (worlds without contact points; narrow
        kernels with contacts up to three rounds of workgroups, or at any batch size with auto-reset on)."""
        return bool(lib().tds_hip_step_many_is_loop(self.h, int(n_steps)))

    def set_graph_chains(self, chains: int):
        """Environment chains of the step_many graphs (0: library default); see tds_hip_step_many in tds_hip.h."""
        _check(lib().tds_hip_set_graph_chains(self.h, int(chains)))

    def tune_step_many(self, actions, probe_steps: int = 64, obs=None) -> int:
        """Measure 1, 2 and 3 chains on ``probe_steps`` steps each (the simulation ADVANCES by 6 x probe_steps steps),
        keep the fastest for later step_many calls and return it."""
        ap, nb, op = self._many_args(actions, obs)
        c = C.c_int()
        _check(lib().tds_hip_step_many_tune(self.h, ap, nb, int(probe_steps), op, C.byref(c)))
        return c.value

    def _many_args(self, actions, obs):
        ap, nb, op = None, 1, None
        if actions is not None:
            assert actions.is_cuda and actions.dtype == self.torch_dtype and actions.is_contiguous()
            assert actions.dim() == 3 and tuple(actions.shape[1:]) == (self.num_envs, self.model.action_dim)
            ap, nb = C.c_void_p(actions.data_ptr()), int(actions.shape[0])
        if obs is not None:
            assert obs.is_cuda and obs.dtype == self.torch_dtype and obs.is_contiguous()
            assert tuple(obs.shape) == (self.num_envs, self.obs_dim + 2)
            op = C.c_void_p(obs.data_ptr())
        return ap, nb, op

    def sync(self):
        _check(lib().tds_hip_sync(self.h))

    # -- the hot path -----------------------------------------------------------------------
    def forward_zero(self, x, y=None):
        """y = f(x) on device tensors [N, input_dim] -> [N, output_dim] (async)."""
        import torch

        assert x.is_cuda and x.dtype == self.torch_dtype and x.is_contiguous()
        assert tuple(x.shape) == (self.num_envs, self.input_dim)
        if y is None:
            y = torch.empty((self.num_envs, self.output_dim), dtype=self.torch_dtype, device=x.device)
        assert y.is_cuda and y.dtype == self.torch_dtype and y.is_contiguous()
        _check(lib().tds_hip_forward_zero_device(self.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
        return y

    def step(self, actions=None, substeps: int = 1, obs=None):
        """Closed-loop step on the resident records (async): x[:, act] <- actions, y = f(x),
        x[:, :nq+nd] <- y[:, :nq+nd].  ``obs`` (optional) [N, obs_dim+2] receives
        [observation | reward | done] from the same launch."""
        ap = None
        if actions is not None:
            assert actions.is_cuda and actions.dtype == self.torch_dtype and actions.is_contiguous()
            assert tuple(actions.shape) == (self.num_envs, self.model.action_dim)
            ap = C.c_void_p(actions.data_ptr())
        op = None
        if obs is not None:
            assert obs.is_cuda and obs.dtype == self.torch_dtype and obs.is_contiguous()
            assert tuple(obs.shape) == (self.num_envs, self.obs_dim + 2)
            op = C.c_void_p(obs.data_ptr())
        _check(lib().tds_hip_step_obs(self.h, ap, int(substeps), op))

    def set_auto_reset(self, enable: bool, seed: int = 0):
        """Reset (+ settle) environments whose step ends with done inside the step launch."""
        _check(lib().tds_hip_set_auto_reset(self.h, 1 if enable else 0, C.c_ulonglong(seed & (2 ** 64 - 1))))

    def reset(self, mask=None, obs=None):
        """Re-initialise + settle the environments selected by ``mask`` (uint8 [N] device tensor,
        None = all) on device; optionally write their observation into ``obs``."""
        mp = None
        if mask is not None:
            import torch

            assert mask.is_cuda and mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == self.num_envs
            mp = C.c_void_p(mask.data_ptr())
        op = None
        if obs is not None:
            assert obs.is_cuda and obs.dtype == self.torch_dtype and obs.is_contiguous()
            assert tuple(obs.shape) == (self.num_envs, self.obs_dim + 2)
            op = C.c_void_p(obs.data_ptr())
        _check(lib().tds_hip_reset(self.h, mp, op))

    def set_policy_network(self, layer_sizes=None, activations=None, use_bias=None):
        """the policy NETWORK of the rollouts (tds_hip_set_policy_network; the reference's NeuralNetworkSpecification):
        layer_sizes incl. the input (obs_dim) and output (action_dim) layers, activations[i - 1] (TDS_NN_ACT_*: -1
        identity, 0 tanh, 1 sin, 2 relu, 3 soft_relu, 4 elu, 5 sigmoid, 6 softsign) for layer i >= 1, use_bias per layer;
        None restores the default linear policy.  Returns the number of parameters per environment."""
        if layer_sizes is None:
            _check(lib().tds_hip_set_policy_network(self.h, 0, None, None, None))
        else:
            n = len(layer_sizes)
            assert len(activations) == n - 1 and len(use_bias) == n
            arr = lambda v: (C.c_int * len(v))(*[int(a) for a in v])
            lib().tds_hip_set_policy_network.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
            _check(lib().tds_hip_set_policy_network(self.h, n, arr(layer_sizes), arr(activations), arr(use_bias)))
        return self.policy_num_parameters

    @property
    def policy_num_parameters(self) -> int:
        lib().tds_hip_policy_num_parameters.argtypes = [C.c_void_p]
        return int(lib().tds_hip_policy_num_parameters(self.h))

    def rollout_ex(self, policy, n_steps: int, shift: float = 0.0, first_obs_raw: bool = False, stats=None,
                   want_traj: bool = False):
        """rollout + the by-products of Worker::rollouts (tds_hip_rollout_ex): ``stats`` [N, obs_dim, 3] device tensor
        of (count, mean, S) updated in place (None: skipped), trajectories [N, n_steps, output_dim] + lengths [N]
        when ``want_traj``.  Returns (return_sum, steps, traj, traj_len)."""
        import torch

        adim, od = self.model.action_dim, self.obs_dim
        assert policy.is_cuda and policy.dtype == self.torch_dtype and policy.is_contiguous()
        ret = torch.zeros(self.num_envs, dtype=self.torch_dtype, device=policy.device)
        steps = torch.zeros(self.num_envs, dtype=torch.int32, device=policy.device)
        sp = None
        if stats is not None:
            assert stats.is_cuda and stats.dtype == self.torch_dtype and stats.is_contiguous()
            assert tuple(stats.shape) == (self.num_envs, od, 3)
            sp = C.c_void_p(stats.data_ptr())
        traj = tlen = None
        tp = lp = None
        if want_traj:
            traj = torch.zeros((self.num_envs, n_steps, self.output_dim), dtype=self.torch_dtype, device=policy.device)
            tlen = torch.zeros(self.num_envs, dtype=torch.int32, device=policy.device)
            tp, lp = C.c_void_p(traj.data_ptr()), C.c_void_p(tlen.data_ptr())
        _check(lib().tds_hip_rollout_ex(self.h, C.c_void_p(policy.data_ptr()), int(n_steps), C.c_double(shift),
                                        1 if first_obs_raw else 0, C.c_void_p(ret.data_ptr()),
                                        C.c_void_p(steps.data_ptr()), None, sp, tp, lp))
        return ret, steps, traj, tlen

    def rollout(self, policy, n_steps: int, shift: float = 0.0, first_obs_raw: bool = False, obs=None, mode=None):
        """n_steps of { action = W obs + b (per-environment linear policy); step; reward/done } on device
        (tds_hip_rollout): in ONE launch, or — from two wavefronts per SIMD on, without auto-reset — as one
        straight-line step launch per step with a small policy + bookkeeping kernel in between
        (mode "per_step" / "single" forces either).  ``policy``: [N, action_dim*obs_dim + action_dim] device
        tensor in NeuralNetwork parameter order.  Returns (return_sum [N], steps [N] int32) device tensors."""
        import torch

        adim, od = self.model.action_dim, self.obs_dim
        assert policy.is_cuda and policy.dtype == self.torch_dtype and policy.is_contiguous()
        assert tuple(policy.shape) == (self.num_envs, self.policy_num_parameters)
        ret = torch.zeros(self.num_envs, dtype=self.torch_dtype, device=policy.device)
        steps = torch.zeros(self.num_envs, dtype=torch.int32, device=policy.device)
        op = None
        if obs is not None:
            assert obs.is_cuda and obs.dtype == self.torch_dtype and obs.is_contiguous()
            assert tuple(obs.shape) == (self.num_envs, od + 2)
            op = C.c_void_p(obs.data_ptr())
        _check(lib().tds_hip_rollout(self.h, C.c_void_p(policy.data_ptr()), int(n_steps), C.c_double(shift),
                                     (1 if first_obs_raw else 0) | {None: 0, "per_step": 2, "single": 4}[mode],
                                     C.c_void_p(ret.data_ptr()),
                                     C.c_void_p(steps.data_ptr()), op))
        return ret, steps

    @property
    def obs_dim(self) -> int:
        return self.model.dof_q + self.model.dof_qd

    def forward_zero_host(self, x_np):
        """Blocking host-buffer call with the reference's <model>_forward_zero semantics."""
        import numpy as np

        x_np = np.ascontiguousarray(x_np, dtype=np.float64).reshape(-1, self.input_dim)
        y_np = np.zeros((x_np.shape[0], self.output_dim), dtype=np.float64)
        _check(lib().tds_hip_forward_zero_host(self.h, x_np.shape[0], x_np.ctypes.data, y_np.ctypes.data))
        return y_np

    def set_timing(self, on: bool):
        _check(lib().tds_hip_set_timing(self.h, 1 if on else 0))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _check(lib().tds_hip_last_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

    PHASES = ["A load+PD", "B jcalc", "C kinematics sweep", "I narrowphase + visuals + D inertias",
              "E composite inertia + bias force sweep", "G mass matrix rows", "H LDLt",
              "F forward dynamics solve", "(barrier)",
              "J jacobian rows", "K row solves", "L PGS", "M/N pack"]

    def profile_phases(self):
        """Shader-clock cycles spent by workgroup 0 in each phase of one step (diagnostic)."""
        buf = (C.c_longlong * 14)()
        _check(lib().tds_hip_profile_phases(self.h, buf, 14))
        st = list(buf)
        return {name: st[i + 1] - st[i] for i, name in enumerate(self.PHASES)}

    def profile_phases_two_waves(self):
        """Raw stamp timelines (shader-clock cycles since the main wavefront's first stamp) of both wavefronts of
        workgroup 0 in the two-wavefront form; None when that form does not serve this grid."""
        nb = (self.num_envs + self.kernel_info()["envs_per_block"] - 1) // self.kernel_info()["envs_per_block"]
        buf = (C.c_longlong * (28 + 2 * nb))()
        _check(lib().tds_hip_profile_phases(self.h, buf, 28 + 2 * nb))
        st = list(buf)
        if st[14] == 0:
            return None
        t0 = st[0]
        # 23, 24: 100 MHz wall clock at workgroup 0's first / last stamp; 25, 26: shader clock at the LAST workgroup's
        # first / last stamp; 27: wall clock at its last stamp
        extra = dict(wg0_wall_us=(st[24] - st[23]) / 100.0, last_wg_start=st[25] - t0, last_wg_end=st[26] - t0,
                     last_wg_end_wall_us=(st[27] - st[23]) / 100.0,
                     wg_start_us=[(v - st[23]) / 100.0 for v in st[28::2]], wg_end_us=[(v - st[23]) / 100.0 for v in st[29::2]])
        return [v - t0 for v in st[:14]], [v - t0 for v in st[14:23]], extra

    def single_step_kernel(self):
        """which kernel a plain single step runs: ("general" | "quad16" | "oct8" | "chain8", lanes per environment, LDS bytes per environment)"""
        a, b = C.c_int(), C.c_int()
        lib().tds_hip_single_step_kernel.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        k = lib().tds_hip_single_step_kernel(self.h, C.byref(a), C.byref(b))
        return {1: "quad16", 2: "oct8", 3: "chain8"}.get(k, "general"), a.value, b.value

    def kernel_info(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _check(lib().tds_hip_kernel_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(lds_bytes_per_env=a.value, lanes_per_env=b.value, envs_per_block=c.value)


class HipShard:
    """This rank's shard of a global batch of environments + the per-policy-step RCCL all-gather of the
    [obs | reward | done] records (tds_hip_shard_*).  ``sim`` is the shard's HipSim (state upload, reset,
    auto-reset ... as usual); ``step(actions)`` steps the shard and submits the exchange on the library's
    communication stream; ``gathered()`` returns the most recently exchanged records as a zero-copy tensor
    [world, block, n_local, obs_dim + 2] in the wire dtype (the current torch stream waits for the exchange)."""

    def __init__(self, m: _model.Model, global_envs: int, rank: int = 0, world: int = 1, device: int = 0,
                 dtype: str = "f64", unique_id: bytes | None = None, wire_dtype: str = "f32", block: int = 1,
                 options: dict | None = None):
        import torch

        if not torch.cuda.is_available():
            raise TdsHipError("no HIP device visible (the HIP path has no CPU fallback)")
        self._model = m.copy()
        h = C.c_void_p()
        idbuf = None
        if unique_id is not None:
            assert len(unique_id) == 128
            idbuf = C.create_string_buffer(bytes(unique_id), 128)
        wire = _model.TDS_DTYPE_F64 if wire_dtype in ("f64", "float64") else _model.TDS_DTYPE_F32
        _check(lib().tds_hip_shard_create(C.byref(self._model), int(global_envs), int(rank), int(world), int(device),
                                          dtype_code(dtype), idbuf, wire, C.byref(h)))
        self.h = h
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self.n_local = lib().tds_hip_shard_local_envs(self.h)
        self.wire_torch_dtype = torch.float64 if lib().tds_hip_shard_wire_bytes(self.h) == 8 else torch.float32
        self.sim = HipSim(m, self.n_local, device=device, dtype=dtype,
                          _handle=C.c_void_p(lib().tds_hip_shard_sim(self.h)), _owner=self)
        self.block = 1
        for k, v in (options or {}).items():  # (run-time options of the shard live in its sim handle)
            self.sim.set_option(k, v)
        if block != 1:
            self.set_block(block)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(lib().tds_hip_shard_unique_id(buf))
        return bytes(buf.raw)

    @staticmethod
    def rccl_version() -> int:
        return int(lib().tds_hip_shard_rccl_version())

    def set_block(self, steps_per_exchange: int):
        _check(lib().tds_hip_shard_set_block(self.h, int(steps_per_exchange)))
        self.block = int(steps_per_exchange)

    def step(self, actions=None, substeps: int = 1):
        ap = None
        if actions is not None:
            assert actions.is_cuda and actions.dtype == self.sim.torch_dtype and actions.is_contiguous()
            assert tuple(actions.shape) == (self.n_local, self.sim.model.action_dim)
            ap = C.c_void_p(actions.data_ptr())
        _check(lib().tds_hip_shard_step(self.h, ap, int(substeps)))

    def step_many(self, actions, n_steps: int, first_block: int = 0, prepare_only: bool = False):
        """n_steps steps + their exchanges as one hipGraph launch; ``actions`` [B, n_local, action_dim] or None.
        prepare_only: capture + instantiate the graph, run nothing."""
        ap, nb = None, 1
        if actions is not None:
            assert actions.is_cuda and actions.dtype == self.sim.torch_dtype and actions.is_contiguous()
            assert actions.dim() == 3 and tuple(actions.shape[1:]) == (self.n_local, self.sim.model.action_dim)
            ap, nb = C.c_void_p(actions.data_ptr()), int(actions.shape[0])
        f = lib().tds_hip_shard_step_many_prepare if prepare_only else lib().tds_hip_shard_step_many
        _check(f(self.h, ap, nb, int(first_block), int(n_steps)))

    def flush(self):
        _check(lib().tds_hip_shard_flush(self.h))

    EXCHANGE_FORMS = {0: "none", 1: "rccl_per_step", 2: "rccl_group_after_launch", 3: "rccl_per_slot", 4: "peer_stores", 5: "peer_copy"}

    def exchange_form(self) -> str:
        """which exchange the most recent step / step_many ran (tds_hip_shard_exchange_form)"""
        lib().tds_hip_shard_exchange_form.argtypes = [C.c_void_p]
        return self.EXCHANGE_FORMS.get(int(lib().tds_hip_shard_exchange_form(self.h)), "?")

    def peer_count(self) -> int:
        """ranks this shard stores its records to under the peer-store exchange; -1: not in use"""
        lib().tds_hip_shard_peer_count.argtypes = [C.c_void_p]
        return int(lib().tds_hip_shard_peer_count(self.h))

    def gathered(self):
        import torch

        ptr, blk = C.c_void_p(), C.c_int()
        st = torch.cuda.current_stream(self.device)
        _check(lib().tds_hip_shard_gathered(self.h, C.c_void_p(st.cuda_stream), C.byref(ptr), C.byref(blk)))
        shape = (self.world, blk.value, self.n_local, self.sim.obs_dim + 2)

        class _Holder:
            pass

        hld = _Holder()
        hld.__cuda_array_interface__ = {
            "shape": shape, "typestr": "<f8" if self.wire_torch_dtype == torch.float64 else "<f4",
            "data": (int(ptr.value), False), "version": 2, "strides": None,
        }
        t = torch.as_tensor(hld, device=f"cuda:{self.device}")
        t._tds_owner = self
        return t

    def gathered_step(self, steps_back: int):
        """ring exchange: gathered records [world, n_local, obs_dim + 2] of the step ``steps_back`` before the most recently
        submitted one, while it is still in the ring (tds_hip_shard_gathered_step)"""
        import torch

        ptr = C.c_void_p()
        st = torch.cuda.current_stream(self.device)
        lib().tds_hip_shard_gathered_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        _check(lib().tds_hip_shard_gathered_step(self.h, int(steps_back), C.c_void_p(st.cuda_stream), C.byref(ptr)))
        return wrap_device_pointer(ptr.value, (self.world, self.n_local, self.sim.obs_dim + 2), self.wire_torch_dtype,
                                   self.device, owner=self)

    def close(self):
        if getattr(self, "h", None):
            self.sim.h = None
            lib().tds_hip_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RigidBodySim:
    """N independent worlds of free rigid bodies (spheres, planes) on one GPU — World::step of the
    reference for tds::RigidBody objects.  ``state`` is a zero-copy torch view
    [N, num_bodies, 13]: position | quaternion xyzw | linear velocity | angular velocity."""

    def __init__(self, m: _model.RbModel, num_worlds: int, device: int = 0, dtype: str = "f64"):
        import torch

        if not torch.cuda.is_available():
            raise TdsHipError("no HIP device visible (the HIP path has no CPU fallback)")
        self.model, self.num_worlds, self.device = m, int(num_worlds), int(device)
        self.dtype = _model.TDS_DTYPE_F64 if dtype in ("f64", "float64") else _model.TDS_DTYPE_F32
        self.torch_dtype = torch.float64 if self.dtype == _model.TDS_DTYPE_F64 else torch.float32
        h = C.c_void_p()
        rc = lib().tds_rb_create(C.byref(m), self.num_worlds, self.device, self.dtype, C.byref(h))
        if rc != TDS_OK:
            raise TdsHipError(f"tds_rb_create error {rc}: {lib().tds_rb_last_error().decode()}")
        self.h = h
        lib().tds_rb_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self.state = HipSim._wrap(self, lib().tds_rb_state_device(self.h),
                                  (self.num_worlds, m.num_bodies, _model.TDS_RB_STATE))

    def step(self, steps: int = 1):
        rc = lib().tds_rb_step(self.h, int(steps))
        if rc != TDS_OK:
            raise TdsHipError(f"tds_rb_step error {rc}: {lib().tds_rb_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None):
            lib().tds_rb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
