"""ctypes binding of the CPU lock-step emulator of the limb kernel (tests/limbemu/limbemu.cc) -- TEST INFRASTRUCTURE ONLY."""
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
        so = os.path.join(_DIR, "liblimbemu.so")
        srcs = [os.path.join(_DIR, "limbemu.cc")] + [os.path.join(_DIR, "..", "..", "mujoco_mpc_amd", "csrc", f) for f in ("limb_step.h", "limb_model.h", "pair_cull.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-o", so, srcs[0]])
        L = C.CDLL(so)
        L.limbemu_check.restype = C.c_char_p
        L.limbemu_check.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask)]
        L.limbemu_forward.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, c_f64p, c_f64p, C.c_int, c_f64p]
        L.limbemu_rollout.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      c_f64p, c_f64p, C.POINTER(MjpcxNoiseSpec), c_f64p, C.c_int] + [c_f64p] * 7 + [c_i32p, c_f64p, c_i32p, c_i32p]
        _LIB = L
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def check(pm, pt):
    return lib().limbemu_check(pm.ptr, pt.ptr).decode()


def forward(pm, pt, state, time, mocap, ctrl, warm=None, precision=64):
    nv, nr = pm.struct.nv, pt.struct.num_residual
    out = np.zeros(3 * nv + nv * nv + 3 + nr + 4 + nv)
    w = None if warm is None else as_f64p(_f(warm))
    fl = lib().limbemu_forward(pm.ptr, pt.ptr, as_f64p(_f(state)), float(time), as_f64p(_f(mocap)), as_f64p(_f(ctrl)), w, precision, as_f64p(out))
    o = 3 * nv + nv * nv
    return dict(flags=fl, qacc=out[0:nv], qfrc_smooth=out[nv:2 * nv], qfrc_constraint=out[2 * nv:3 * nv], M=out[3 * nv:o].reshape(nv, nv), com=out[o:o + 3],
                residual=out[o + 3:o + 3 + nr], cost=out[o + 3 + nr], nx=int(out[o + 3 + nr + 2]), iters=int(out[o + 3 + nr + 3]), qacc_smooth=out[o + 3 + nr + 4:])


def rollout(pm, pt, state, time, mocap, N, H, P, interp, node_times, node_values=None, noise=None, nominal=None, precision=64):
    m = pm.struct
    ds, nu, nr, ntr = m.nq + m.nv, m.nu, pt.struct.num_residual, pt.struct.num_trace
    out = dict(states=np.zeros((N, H, ds)), actions=np.zeros((N, H, nu)), times=np.zeros((N, H)), residual=np.zeros((N, H, nr)),
               costs=np.zeros((N, H)), trace=np.zeros((N, H, 3 * ntr)), total_return=np.zeros(N), failure=np.zeros(N, np.int32),
               nodes=np.zeros((N, P, nu)), flags=np.zeros(N, np.int32), iters=np.zeros(N, np.int32))
    nv = None if node_values is None else as_f64p(_f(node_values))
    nom = as_f64p(_f(nominal if nominal is not None else np.zeros((P, nu))))
    rc = lib().limbemu_rollout(pm.ptr, pt.ptr, as_f64p(_f(state)), float(time), as_f64p(_f(mocap)), N, H, P, interp, as_f64p(_f(node_times)), nv,
                               None if noise is None else C.byref(noise), nom, precision,
                               as_f64p(out["states"]), as_f64p(out["actions"]), as_f64p(out["times"]), as_f64p(out["residual"]), as_f64p(out["costs"]),
                               as_f64p(out["trace"]), as_f64p(out["total_return"]), as_i32p(out["failure"]), as_f64p(out["nodes"]), as_i32p(out["flags"]),
                               as_i32p(out["iters"]))
    if rc != 0:
        raise RuntimeError("limb kernel does not cover this model / task: " + check(pm, pt))
    return out
