"""ctypes binding of the CPU lock-step emulator of the quad kernel (tests/quademu/quademu.cc) -- TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from mujoco_mpc_amd.cstructs import MjpcxModel, MjpcxNoiseSpec, MjpcxTask, as_f64p, as_i32p, c_f64p, c_i32p

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_DIR, "libquademu.so")
        srcs = [os.path.join(_DIR, "quademu.cc")] + [os.path.join(_DIR, "..", "..", "mujoco_mpc_amd", "csrc", f) for f in ("quad_step.h", "quad_model.h", "solid_pairs.h", "pair_cull.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-o", so, srcs[0]])
        L = C.CDLL(so)
        L.quademu_check.restype = C.c_char_p
        L.quademu_check.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask)]
        L.quademu_forward.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, c_f64p, c_f64p, c_f64p]
        L.quademu_rollout.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      c_f64p, c_f64p, C.POINTER(MjpcxNoiseSpec), c_f64p] + [c_f64p] * 7 + [c_i32p, c_f64p, c_i32p]
        L.quademu_rollout_feedback.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_int] + [c_f64p] * 6 + [c_f64p] * 7 + [c_i32p, c_i32p]
        _LIB = L
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def plan(masks):
    """quad_step.h make_plan on the four legs' contact masks -> (nslots, x[3], eslot[4], cyclic), the same in the four lanes"""
    import numpy as np
    out = np.zeros((4, 9), np.int32)
    m = np.ascontiguousarray(masks, np.int32)
    lib().quademu_plan(m.ctypes.data_as(C.POINTER(C.c_int)), out.ctypes.data_as(C.POINTER(C.c_int)))
    assert (out == out[0]).all(), out
    return int(out[0, 0]), out[0, 1:4].tolist(), out[0, 4:8].tolist(), bool(out[0, 8])


def check(pm, pt):
    return lib().quademu_check(pm.ptr, pt.ptr).decode()


def forward(pm, pt, state, time, mocap, ctrl, warm=None):
    out = np.zeros(54 + 324 + 3 + 42 + 3)
    w = None if warm is None else as_f64p(_f(warm))
    fl = lib().quademu_forward(pm.ptr, pt.ptr, as_f64p(_f(state)), float(time), as_f64p(_f(mocap)), as_f64p(_f(ctrl)), w, as_f64p(out))
    return dict(flags=fl, qacc=out[0:18], qfrc_smooth=out[18:36], qfrc_constraint=out[36:54], M=out[54:378].reshape(18, 18), com=out[378:381],
                residual=out[381:423], cost=out[423], iters=int(out[425]))


def rollout(pm, pt, state, time, mocap, N, H, P, interp, node_times, node_values=None, noise=None, nominal=None):
    m = pm.struct
    ds, nu, nr, ntr = m.nq + m.nv, m.nu, pt.struct.num_residual, pt.struct.num_trace
    out = dict(states=np.zeros((N, H, ds)), actions=np.zeros((N, H, nu)), times=np.zeros((N, H)), residual=np.zeros((N, H, nr)),
               costs=np.zeros((N, H)), trace=np.zeros((N, H, 3 * ntr)), total_return=np.zeros(N), failure=np.zeros(N, np.int32),
               nodes=np.zeros((N, P, nu)), flags=np.zeros(N, np.int32))
    nv = None if node_values is None else as_f64p(_f(node_values))
    nom = as_f64p(_f(nominal if nominal is not None else np.zeros((P, nu))))
    rc = lib().quademu_rollout(pm.ptr, pt.ptr, as_f64p(_f(state)), float(time), as_f64p(_f(mocap)), N, H, P, interp, as_f64p(_f(node_times)), nv,
                               None if noise is None else C.byref(noise), nom,
                               as_f64p(out["states"]), as_f64p(out["actions"]), as_f64p(out["times"]), as_f64p(out["residual"]), as_f64p(out["costs"]),
                               as_f64p(out["trace"]), as_f64p(out["total_return"]), as_i32p(out["failure"]), as_f64p(out["nodes"]), as_i32p(out["flags"]))
    if rc != 0:
        raise RuntimeError("quad kernel does not cover this model / task: " + check(pm, pt))
    return out


def rollout_feedback(pm, pt, state, time, mocap, N, H, mode, representation, use_state, times, states, actions, gains, improvement, alpha):
    """N rollouts under the iLQG feedback policy (the arguments of capi.Context.rollout_feedback / pyoracle.rollout_feedback)"""
    m = pm.struct
    ds, nu, nr, ntr = m.nq + m.nv, m.nu, pt.struct.num_residual, pt.struct.num_trace
    Tn = len(times)
    out = dict(states=np.zeros((N, H, ds)), actions=np.zeros((N, H, nu)), times=np.zeros((N, H)), residual=np.zeros((N, H, nr)),
               costs=np.zeros((N, H)), trace=np.zeros((N, H, 3 * ntr)), total_return=np.zeros(N), failure=np.zeros(N, np.int32), flags=np.zeros(N, np.int32))
    keep = [_f(times), _f(states), _f(actions), _f(gains), _f(improvement), _f(alpha)]
    rc = lib().quademu_rollout_feedback(pm.ptr, pt.ptr, as_f64p(_f(state)), float(time), as_f64p(_f(mocap)), N, H, mode, representation, int(use_state), Tn,
                                        *[as_f64p(k) for k in keep],
                                        as_f64p(out["states"]), as_f64p(out["actions"]), as_f64p(out["times"]), as_f64p(out["residual"]), as_f64p(out["costs"]),
                                        as_f64p(out["trace"]), as_f64p(out["total_return"]), as_i32p(out["failure"]), as_i32p(out["flags"]))
    if rc != 0:
        raise RuntimeError("quad kernel does not cover this model / task: " + check(pm, pt))
    return out
