"""ctypes mirrors of the structs in include/mjpcx.h, and FlatModel -> mjpcx_model packing."""
from __future__ import annotations

import ctypes as C

import numpy as np

c_i32p = C.POINTER(C.c_int32)
c_f64p = C.POINTER(C.c_double)

_MODEL_INT_ARRAYS = [
    "body_parentid", "body_rootid", "body_jntnum", "body_jntadr", "body_dofnum", "body_dofadr", "body_mocapid",
]
# order must match include/mjpcx.h exactly
_MODEL_FIELDS = (
    [(n, C.c_int32) for n in ("nq", "nv", "nu", "na", "nbody", "njnt", "nsite", "nmocap", "nuserdata")]
    + [("timestep", C.c_double), ("gravity", C.c_double * 3), ("integrator", C.c_int32),
       ("disableflags", C.c_int32), ("solver_iterations", C.c_int32), ("solver_tolerance", C.c_double),
       ("meaninertia", C.c_double)]
    + [(n, c_i32p) for n in _MODEL_INT_ARRAYS]
    + [(n, c_f64p) for n in ("body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_inertia")]
    + [(n, c_i32p) for n in ("jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited")]
    + [(n, c_f64p) for n in ("jnt_pos", "jnt_axis", "jnt_stiffness", "jnt_range", "jnt_margin", "jnt_solref", "jnt_solimp")]
    + [(n, c_i32p) for n in ("dof_bodyid", "dof_jntid", "dof_parentid")]
    + [(n, c_f64p) for n in ("dof_armature", "dof_damping", "dof_frictionloss", "dof_invweight0")]
    + [(n, c_f64p) for n in ("qpos0", "qpos_spring")]
    + [("site_bodyid", c_i32p), ("site_pos", c_f64p), ("site_quat", c_f64p)]
    + [(n, c_i32p) for n in ("actuator_trnid", "actuator_gaintype", "actuator_biastype",
                             "actuator_ctrllimited", "actuator_forcelimited")]
    + [(n, c_f64p) for n in ("actuator_gear", "actuator_gainprm", "actuator_biasprm",
                             "actuator_ctrlrange", "actuator_forcerange")]
    + [("ngeom", C.c_int32), ("nkey", C.c_int32), ("cone", C.c_int32), ("impratio", C.c_double)]
    + [(n, c_i32p) for n in ("geom_type", "geom_bodyid", "geom_contype", "geom_conaffinity", "geom_condim",
                             "geom_priority", "geom_group")]
    + [(n, c_f64p) for n in ("geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_solref", "geom_solimp",
                             "geom_margin", "geom_gap", "geom_solmix", "body_invweight0", "body_subtreemass",
                             "dof_solref", "dof_solimp", "key_qpos")]
    + [(n, C.c_int32) for n in ("ntendon", "nwrap", "nexclude")]
    + [(n, c_i32p) for n in ("tendon_adr", "tendon_num", "tendon_limited", "wrap_objid")]
    + [(n, c_f64p) for n in ("wrap_prm", "tendon_range", "tendon_margin", "tendon_solref_lim", "tendon_solimp_lim",
                             "tendon_invweight0")]
    + [("exclude_signature", c_i32p), ("body_weldid", c_i32p), ("key_mpos", c_f64p)]
)


class MjpcxModel(C.Structure):
    _fields_ = _MODEL_FIELDS


class MjpcxTask(C.Structure):
    _fields_ = [
        ("residual_id", C.c_int32), ("num_residual", C.c_int32), ("num_term", C.c_int32),
        ("num_trace", C.c_int32), ("num_parameter", C.c_int32),
        ("dim_norm_residual", c_i32p), ("norm", c_i32p), ("num_norm_parameter", c_i32p),
        ("weight", c_f64p), ("norm_parameter", c_f64p), ("parameters", c_f64p),
        ("trace_site", c_i32p), ("risk", C.c_double),
        ("num_residual_int", C.c_int32), ("num_residual_real", C.c_int32),
        ("residual_int", c_i32p), ("residual_real", c_f64p),
    ]


class MjpcxTrajView(C.Structure):
    _fields_ = [
        ("horizon", C.c_int32), ("states", c_f64p), ("actions", c_f64p), ("times", c_f64p),
        ("residual", c_f64p), ("costs", c_f64p), ("trace", c_f64p),
        ("total_return", C.c_double), ("failure", C.c_int32),
    ]


class MjpcxNoiseSpec(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("iteration", C.c_uint32), ("mode", C.c_int32),
        ("candidate_offset", C.c_int32), ("nominal_candidate", C.c_int32), ("explore_count", C.c_int32),
        ("std0", C.c_double), ("std1", C.c_double), ("param_variance", c_f64p),
    ]


def as_f64p(a):
    return a.ctypes.data_as(c_f64p)


def as_i32p(a):
    return a.ctypes.data_as(c_i32p)


class PackedModel:
    """Owns contiguous copies of the arrays an MjpcxModel struct points into."""

    def __init__(self, fm, timestep=None, integrator=None, differentiable=False):
        self.fm = fm
        self.keep = {}
        m = MjpcxModel()
        for name, ctype in _MODEL_FIELDS:
            if ctype is C.c_int32:
                setattr(m, name, int(fm.scalars[name]))
            elif ctype is C.c_double:
                setattr(m, name, float(fm.scalars[name]))
            elif ctype is c_i32p:
                arr = np.ascontiguousarray(fm.arrays[name], dtype=np.int32).reshape(-1)
                if arr.size == 0:
                    arr = np.zeros(1, np.int32)
                self.keep[name] = arr
                setattr(m, name, as_i32p(arr))
            elif ctype is c_f64p:
                arr = np.array(fm.arrays[name], dtype=np.float64).reshape(-1)   # private copy
                if differentiable and name in ("jnt_solimp", "geom_solimp") and arr.size:
                    arr[0::5] = 0.0   # MakeDifferentiable, mjpc/utilities.cc:60-75: solimp[0] = 0
                if arr.size == 0:
                    arr = np.zeros(1, np.float64)
                self.keep[name] = arr
                setattr(m, name, as_f64p(arr))
            else:  # gravity
                g = fm.scalars["gravity"]
                m.gravity[0], m.gravity[1], m.gravity[2] = float(g[0]), float(g[1]), float(g[2])
        # Agent::PlanIteration overrides the planning copy's timestep / integrator
        # with agent_timestep / agent_integrator (mjpc/agent.cc:288-291)
        if timestep is not None:
            m.timestep = float(timestep)
        if integrator is not None:
            m.integrator = int(integrator)
        self.struct = m

    @property
    def ptr(self):
        return C.byref(self.struct)


class PackedTask:
    def __init__(self, spec: dict):
        """spec keys = mjpcx_task fields (python lists / numpy arrays / scalars)."""
        self.spec = spec
        t = MjpcxTask()
        self.keep = {}
        for name, ctype in MjpcxTask._fields_:
            v = spec[name]
            if ctype is c_i32p:
                arr = np.ascontiguousarray(v, dtype=np.int32).reshape(-1)
                if arr.size == 0:
                    arr = np.zeros(1, np.int32)
                self.keep[name] = arr
                setattr(t, name, as_i32p(arr))
            elif ctype is c_f64p:
                arr = np.ascontiguousarray(v, dtype=np.float64).reshape(-1)
                if arr.size == 0:
                    arr = np.zeros(1, np.float64)
                self.keep[name] = arr
                setattr(t, name, as_f64p(arr))
            else:
                setattr(t, name, v)
        self.struct = t

    @property
    def ptr(self):
        return C.byref(self.struct)
