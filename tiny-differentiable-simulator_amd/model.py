"""ctypes mirror of ``tds_model_t`` (include/tds_hip.h) and JSON (de)serialisation.

The blob is the flattened output of TDS's own URDF loader + World set-up
(reference: src/urdf/urdf_to_multi_body.hpp:41-220, examples/environments/
locomotion_contact_simulation.h:88-136).  On a machine where the reference is present it is
produced by ``include/tds_hip_stepper.hpp::flatten_model`` (C++) or ``oracle/gen_golden.py``;
the four models of BASELINE.json's configs are committed under ``models/*.json`` so that the
GPU box (which has no /root/reference) can run them.
"""
from __future__ import annotations

import ctypes as C
import json
import os

TDS_HIP_ABI_VERSION = 5
TDS_MAX_LINKS = 64
TDS_MAX_GEOMS = 32
TDS_MAX_VISUALS = 64
TDS_MAX_ACTIONS = 32
TDS_MAX_DOF = 32
TDS_MAX_CONTACTS = 64
TDS_MAX_BODIES = 4

TDS_STEP_LOCOMOTION = 0
TDS_STEP_TAU = 1
TDS_REWARD_NONE, TDS_REWARD_ANT, TDS_REWARD_LAIKAGO, TDS_REWARD_HUMANOID = 0, 1, 2, 3
TDS_DTYPE_F64 = 0
TDS_DTYPE_F32 = 1
TDS_DTYPE_F64_REC32 = 2  # double arithmetic, float records (include/tds_hip.h)

JOINT_FIXED = -1
JOINT_PRISMATIC_X, JOINT_PRISMATIC_Y, JOINT_PRISMATIC_Z, JOINT_PRISMATIC_AXIS = 0, 1, 2, 3
JOINT_REVOLUTE_X, JOINT_REVOLUTE_Y, JOINT_REVOLUTE_Z, JOINT_REVOLUTE_AXIS = 4, 5, 6, 7
JOINT_SPHERICAL = 8
GEOM_SPHERE, GEOM_PLANE, GEOM_CAPSULE, GEOM_MESH, GEOM_BOX = 0, 1, 2, 3, 4


class Link(C.Structure):
    _fields_ = [
        ("joint_type", C.c_int32),
        ("parent", C.c_int32),
        ("q_index", C.c_int32),
        ("qd_index", C.c_int32),
        ("X_T_rot", C.c_double * 9),
        ("X_T_trans", C.c_double * 3),
        ("S", C.c_double * 6),
        ("mass", C.c_double),
        ("com", C.c_double * 3),
        ("inertia", C.c_double * 9),
        ("stiffness", C.c_double),
        ("damping", C.c_double),
    ]


class Geom(C.Structure):
    _fields_ = [
        ("link", C.c_int32),
        ("type", C.c_int32),
        ("radius", C.c_double),
        ("length", C.c_double),
        ("extents", C.c_double * 3),
        ("X_rot", C.c_double * 9),
        ("X_trans", C.c_double * 3),
    ]


class Visual(C.Structure):
    _fields_ = [
        ("link", C.c_int32),
        ("pad_", C.c_int32),
        ("X_rot", C.c_double * 9),
        ("X_trans", C.c_double * 3),
    ]


class Body(C.Structure):
    """body b >= 1 of a world with several articulated bodies (tds_body_t)"""
    _fields_ = [
        ("first_link", C.c_int32),
        ("first_geom", C.c_int32),
        ("is_floating", C.c_int32),
        ("pad_", C.c_int32),
        ("base_X_world_rot", C.c_double * 9),
        ("base_X_world_trans", C.c_double * 3),
        ("base_mass", C.c_double),
        ("base_com", C.c_double * 3),
        ("base_inertia", C.c_double * 9),
    ]


class Model(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("step_mode", C.c_int32),
        ("num_links", C.c_int32),
        ("dof_q", C.c_int32),
        ("dof_qd", C.c_int32),
        ("is_floating", C.c_int32),
        ("num_geoms", C.c_int32),
        ("num_visuals", C.c_int32),
        ("action_dim", C.c_int32),
        ("pd_start_link", C.c_int32),
        ("has_plane", C.c_int32),
        ("pgs_iterations", C.c_int32),
        ("input_dim", C.c_int32),
        ("output_dim", C.c_int32),
        ("pack_visuals", C.c_int32),
        ("reward_mode", C.c_int32),
        ("dt", C.c_double),
        ("gravity", C.c_double * 3),
        ("base_X_world_rot", C.c_double * 9),
        ("base_X_world_trans", C.c_double * 3),
        ("plane_normal", C.c_double * 3),
        ("plane_constant", C.c_double),
        ("cfm", C.c_double),
        ("erp", C.c_double),
        ("friction", C.c_double),
        ("restitution", C.c_double),
        ("action_limit", C.c_double),
        ("initial_poses", C.c_double * TDS_MAX_ACTIONS),
        ("reset_q", C.c_double * TDS_MAX_DOF),
        ("reset_noise", C.c_double * TDS_MAX_DOF),
        ("settle_steps", C.c_int32),
        ("reset_obs_raw_xy", C.c_int32),
        ("base_mass", C.c_double),
        ("base_com", C.c_double * 3),
        ("base_inertia", C.c_double * 9),
        ("num_bodies", C.c_int32),
        ("pad3_", C.c_int32 * 3),
        ("bodies", Body * TDS_MAX_BODIES),
        ("links", Link * TDS_MAX_LINKS),
        ("geoms", Geom * TDS_MAX_GEOMS),
        ("visuals", Visual * TDS_MAX_VISUALS),
        ("name", C.c_char * 32),
    ]

    # ---- helpers ---------------------------------------------------------------------
    def copy(self) -> "Model":
        m = Model()
        C.memmove(C.byref(m), C.byref(self), C.sizeof(Model))
        return m

    @property
    def num_contacts(self) -> int:
        """Contact points vs. the plane: sphere 1, capsule 2, box 8
        (reference: src/contact_point.hpp:96-198)."""
        n = 0
        for g in range(self.num_geoms):
            t = self.geoms[g].type
            n += {GEOM_SPHERE: 1, GEOM_CAPSULE: 2, GEOM_BOX: 8}.get(t, 0)
        return n if self.has_plane else 0

    def body_table(self):
        """per articulated body of the world: dict(links=(l0, l1), geoms=(g0, g1), q=(q0, q1), qd=(d0, d1),
        tau=(t0, t1), floating) — index ranges into the link / geom tables and the q | qd | tau parts of the records
        (one entry for a single-body model)"""
        B = max(1, self.num_bodies)
        out, q, d, t = [], 0, 0, 0
        for b in range(B):
            l0 = 0 if b == 0 else self.bodies[b].first_link
            l1 = self.bodies[b + 1].first_link if b + 1 < B else self.num_links
            g0 = 0 if b == 0 else self.bodies[b].first_geom
            g1 = self.bodies[b + 1].first_geom if b + 1 < B else self.num_geoms
            fl = bool(self.is_floating if b == 0 else self.bodies[b].is_floating)
            nj = sum(3 if self.links[i].joint_type == JOINT_SPHERICAL else 1
                     for i in range(l0, l1) if self.links[i].joint_type != JOINT_FIXED)
            nsph = sum(1 for i in range(l0, l1) if self.links[i].joint_type == JOINT_SPHERICAL)
            nq, nd = nj + nsph + (7 if fl else 0), nj + (6 if fl else 0)
            out.append(dict(links=(l0, l1), geoms=(g0, g1), q=(q, q + nq), qd=(d, d + nd), tau=(t, t + nj), floating=fl))
            q, d, t = q + nq, d + nd, t + nj
        return out

    def set_soft_contact(self, stiffness: float, damping: float) -> None:
        """cfm/erp from a contact stiffness/damping pair — the only "spring-damper" contact
        the reference has (reference: examples/environments/ant_environment.h:79-92)."""
        dt = self.dt
        self.cfm = 1.0 / (dt * stiffness + damping)
        self.erp = dt * stiffness / (dt * stiffness + damping)


_SCALARS = [
    "abi_version", "step_mode", "num_links", "dof_q", "dof_qd", "is_floating", "num_geoms",
    "num_visuals", "action_dim", "pd_start_link", "has_plane", "pgs_iterations", "input_dim",
    "output_dim", "pack_visuals", "reward_mode", "settle_steps", "dt", "plane_constant", "cfm", "erp", "friction",
    "restitution", "action_limit",
]
_VECTORS = ["gravity", "base_X_world_rot", "base_X_world_trans", "plane_normal"]
_OPTIONAL_VECTORS = ["base_com", "base_inertia"]  # JSON files older than the floating-base fields lack them


def _struct_to_dict(s, skip=()):
    d = {}
    for name, typ in s._fields_:
        if name in skip or name == "pad_":
            continue
        v = getattr(s, name)
        if hasattr(v, "__len__") and not isinstance(v, (bytes, str)):
            d[name] = [float(x) for x in v]
        else:
            d[name] = v
    return d


def _dict_to_struct(d, s):
    for name, typ in s._fields_:
        if name == "pad_" or name not in d:
            continue
        v = d[name]
        if isinstance(v, list):
            arr = getattr(s, name)
            for i, x in enumerate(v):
                arr[i] = x
        else:
            setattr(s, name, v)


def model_to_dict(m: Model) -> dict:
    d = {k: getattr(m, k) for k in _SCALARS}
    for k in _VECTORS:
        d[k] = [float(x) for x in getattr(m, k)]
    d["reset_obs_raw_xy"] = int(m.reset_obs_raw_xy)
    if m.num_bodies > 1:  # multi-body worlds only: the single-body files stay as they are
        d["num_bodies"] = int(m.num_bodies)
        d["bodies"] = [_struct_to_dict(m.bodies[b]) for b in range(1, m.num_bodies)]   # (body 0: the model's own fields)
    d["base_mass"] = float(m.base_mass)
    for k in _OPTIONAL_VECTORS:
        d[k] = [float(x) for x in getattr(m, k)]
    d["name"] = m.name.decode()
    d["initial_poses"] = [float(m.initial_poses[i]) for i in range(m.action_dim)] \
        if m.step_mode == TDS_STEP_LOCOMOTION else []
    d["reset_q"] = [float(m.reset_q[i]) for i in range(m.dof_q)]
    d["reset_noise"] = [float(m.reset_noise[i]) for i in range(m.dof_q)]
    d["links"] = [_struct_to_dict(m.links[i]) for i in range(m.num_links)]
    d["geoms"] = [_struct_to_dict(m.geoms[i]) for i in range(m.num_geoms)]
    d["visuals"] = [_struct_to_dict(m.visuals[i]) for i in range(m.num_visuals)]
    return d


def model_from_dict(d: dict) -> Model:
    m = Model()
    for k in _SCALARS:
        setattr(m, k, d[k])
    for k in _VECTORS:
        arr = getattr(m, k)
        for i, x in enumerate(d[k]):
            arr[i] = x
    m.reset_obs_raw_xy = d.get("reset_obs_raw_xy", 0)
    m.num_bodies = d.get("num_bodies", 0)
    for b, bd in enumerate(d.get("bodies", [])):
        _dict_to_struct(bd, m.bodies[b + 1])
    m.base_mass = d.get("base_mass", 0.0)
    for k in _OPTIONAL_VECTORS:
        arr = getattr(m, k)
        for i, x in enumerate(d.get(k, [])):
            arr[i] = x
    m.name = d["name"].encode()
    for i, x in enumerate(d["initial_poses"]):
        m.initial_poses[i] = x
    for i, x in enumerate(d.get("reset_q", [])):
        m.reset_q[i] = x
    for i, x in enumerate(d.get("reset_noise", [])):
        m.reset_noise[i] = x
    for i, l in enumerate(d["links"]):
        _dict_to_struct(l, m.links[i])
    for i, g in enumerate(d["geoms"]):
        _dict_to_struct(g, m.geoms[i])
    for i, v in enumerate(d["visuals"]):
        _dict_to_struct(v, m.visuals[i])
    # the JSON form is layout-free: fields it lacks stay zero; stamp the blob with the layout we filled
    m.abi_version = TDS_HIP_ABI_VERSION
    return m


def save_model(m: Model, path: str) -> None:
    with open(path, "w") as f:
        json.dump(model_to_dict(m), f, indent=1)
        f.write("\n")


def load_model(name_or_path: str) -> Model:
    """Load a committed model by name ("ant", "laikago", "cartpole", "pendulum5", ...) or
    from an explicit JSON path."""
    path = name_or_path
    if not os.path.exists(path):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models",
                            name_or_path + ".json")
    with open(path) as f:
        return model_from_dict(json.load(f))


def available_models():
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
    return sorted(p[:-5] for p in os.listdir(d) if p.endswith(".json"))


# ---------------------------------------------------------------------------------------------
# free rigid bodies (include/tds_hip.h: tds_rb_body_t / tds_rb_model_t)
# ---------------------------------------------------------------------------------------------
TDS_RB_MAX_BODIES = 16
TDS_RB_STATE = 13


class RbBody(C.Structure):
    _fields_ = [
        ("mass", C.c_double),
        ("geom_type", C.c_int32),
        ("pad_", C.c_int32),
        ("radius", C.c_double),
        ("length", C.c_double),
        ("extents", C.c_double * 3),
        ("plane_normal", C.c_double * 3),
        ("plane_constant", C.c_double),
    ]


class RbModel(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("num_bodies", C.c_int32),
        ("solver_iterations", C.c_int32),
        ("pad_", C.c_int32),
        ("dt", C.c_double),
        ("gravity", C.c_double * 3),
        ("restitution", C.c_double),
        ("friction", C.c_double),
        ("erp", C.c_double),
        ("bodies", RbBody * TDS_RB_MAX_BODIES),
    ]


def make_rb_model(bodies, dt=1.0 / 240.0, gravity=(0.0, 0.0, -9.81), solver_iterations=1,
                  restitution=0.0, friction=0.5, erp=0.1) -> RbModel:
    """bodies: list of dicts {"mass": m, "sphere": radius} | {"mass": m, "capsule": (radius, length)} |
    {"mass": m, "box": (ex, ey, ez)} | {"mass": 0, "plane": (nx, ny, nz, constant)};
    the defaults are the reference's (world.hpp:65-71, rb_constraint_solver.hpp:45)."""
    m = RbModel()
    m.abi_version = TDS_HIP_ABI_VERSION
    m.num_bodies = len(bodies)
    assert 1 <= len(bodies) <= TDS_RB_MAX_BODIES
    m.solver_iterations = solver_iterations
    m.dt = dt
    for k in range(3):
        m.gravity[k] = gravity[k]
    m.restitution, m.friction, m.erp = restitution, friction, erp
    for i, b in enumerate(bodies):
        m.bodies[i].mass = b["mass"]
        if "sphere" in b:
            m.bodies[i].geom_type = GEOM_SPHERE
            m.bodies[i].radius = b["sphere"]
        elif "capsule" in b:
            m.bodies[i].geom_type = GEOM_CAPSULE
            m.bodies[i].radius, m.bodies[i].length = b["capsule"]
        elif "box" in b:
            m.bodies[i].geom_type = GEOM_BOX
            for k, v in enumerate(b["box"]):
                m.bodies[i].extents[k] = v
        else:
            m.bodies[i].geom_type = GEOM_PLANE
            nx, ny, nz, c = b["plane"]
            for k, v in enumerate((nx, ny, nz)):
                m.bodies[i].plane_normal[k] = v
            m.bodies[i].plane_constant = c
    return m
