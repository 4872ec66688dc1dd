"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package (mujoco_mpc_amd/)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from mujoco_mpc_amd.cstructs import (MjpcxModel, MjpcxNoiseSpec, MjpcxTask, MjpcxTrajView, as_f64p, as_i32p,
                                     c_f64p, c_i32p)

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False, fast=False):
    name = "liboracle_fast.so" if fast else "liboracle.so"
    so = os.path.join(_DIR, name)
    # always ask make: it rebuilds only when a source, an .inc or the C-ABI header is newer than the library
    subprocess.check_call(["make", "-C", _DIR, "-s", name] + (["-B"] if force else []))
    return so


_FAST = None


def rollout_batch_fast(pm, pt, state, time, mocap, N, H, P, interp, node_times, node_values, num_threads=1):
    """orollout_batch from the -O3 -march=native build (timing only: bench.py's cpu_baseline). Returns total_return."""
    global _FAST
    if _FAST is None:
        _FAST = C.CDLL(build(fast=True))
        _FAST.orollout_batch.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, c_f64p,
                                         C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p, C.c_int, C.POINTER(OBatchOut)]
    ret, fail = np.zeros(N), np.zeros(N, np.int32)
    o = OBatchOut()
    o.total_return, o.failure = as_f64p(ret), as_i32p(fail)
    nt, nvv = _f(node_times), _f(node_values)
    mc = None if mocap is None else as_f64p(_f(mocap))
    _FAST.orollout_batch(pm.ptr, pt.ptr, as_f64p(_f(state)), float(time), mc, None, N, H, P, interp, as_f64p(nt),
                         as_f64p(nvv), int(num_threads), C.byref(o))
    return ret


class OSpline(C.Structure):
    _fields_ = [("dim", C.c_int), ("interp", C.c_int), ("size", C.c_int), ("cap", C.c_int),
                ("times", c_f64p), ("values", c_f64p)]


class OBatchOut(C.Structure):
    _fields_ = [("total_return", c_f64p), ("failure", c_i32p), ("states", c_f64p), ("actions", c_f64p),
                ("times", c_f64p), ("residual", c_f64p), ("costs", c_f64p), ("trace", c_f64p)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load(build())
    return _LIB


import contextlib


@contextlib.contextmanager
def timing_build():
    """Inside the block every call of this module goes to the -O3 -march=native -flto build of the same sources (the way the reference
    builds its release binaries): bench.py's cpu_baseline legs only, never the parity checker. Handles (Physics) made inside the
    block belong to that library and must not outlive it."""
    global _LIB
    saved = _LIB
    _LIB = _load(build(fast=True))
    try:
        yield
    finally:
        _LIB = saved


def flops_build():
    """liboracle_flops.so: the oracle's sources compiled with an operation-counting `double` (oracle/flopcount.h)."""
    subprocess.check_call(["make", "-C", _DIR, "-s", "liboracle_flops.so"], stdout=subprocess.DEVNULL)
    return os.path.join(_DIR, "liboracle_flops.so")


@contextlib.contextmanager
def counting_flops(out):
    """Inside the block every call of this module goes to the operation-counting build; on exit `out` (a dict) holds the operations
    executed inside it: add (additions and subtractions), mul, div, sqrt, libm (other math-library calls, one each), flop (their sum).
    The flop count of the REFERENCE ALGORITHM as the oracle restates it -- bench.py's roofline.fp64 numerator; never a checker, never timed."""
    global _LIB
    saved = _LIB
    _LIB = _load(flops_build())
    _LIB.oracle_flops_reset()
    try:
        yield
    finally:
        v = (C.c_ulonglong * 5)()
        _LIB.oracle_flops_read(v)
        out.update(add=int(v[0]), mul=int(v[1]), div=int(v[2]), sqrt=int(v[3]), libm=int(v[4]), flop=int(sum(v)))
        _LIB = saved


def _load(path):
    if True:
        L = C.CDLL(path)
        L.odata_new.restype = C.c_void_p
        L.odata_new.argtypes = [C.POINTER(MjpcxModel)]
        L.odata_free.argtypes = [C.c_void_p]
        L.odata_set_state.argtypes = [C.c_void_p, c_f64p, C.c_double, c_f64p, c_f64p]
        L.odata_set_ctrl.argtypes = [C.c_void_p, c_f64p]
        L.o_forward.argtypes = [C.c_void_p]
        L.o_solve_pgs.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.o_solve_pgs.restype = C.c_int
        L.o_step.argtypes = [C.c_void_p]
        L.o_forward_task.argtypes = [C.c_void_p, C.POINTER(MjpcxTask), c_f64p]
        L.o_step_task.argtypes = [C.c_void_p, C.POINTER(MjpcxTask), c_f64p]
        L.odata_warning.argtypes = [C.c_void_p]
        L.odata_get.argtypes = [C.c_void_p, C.c_char_p, c_f64p, C.c_int]
        L.onorm.restype = C.c_double
        L.onorm.argtypes = [c_f64p, c_f64p, c_f64p, c_f64p, C.c_int, C.c_int]
        L.ocost_value.restype = C.c_double
        L.ocost_value.argtypes = [C.POINTER(MjpcxTask), c_f64p]
        L.ocost_terms.argtypes = [C.POINTER(MjpcxTask), c_f64p, c_f64p, C.c_int]
        L.ospline_init.argtypes = [C.POINTER(OSpline), C.c_int, C.c_int]
        L.ospline_free.argtypes = [C.POINTER(OSpline)]
        L.ospline_clear.argtypes = [C.POINTER(OSpline)]
        L.ospline_copy.argtypes = [C.POINTER(OSpline), C.POINTER(OSpline)]
        L.ospline_set_interpolation.argtypes = [C.POINTER(OSpline), C.c_int]
        L.ospline_add_node.argtypes = [C.POINTER(OSpline), C.c_double, c_f64p]
        L.ospline_sample.argtypes = [C.POINTER(OSpline), C.c_double, c_f64p]
        L.ospline_discard_before.argtypes = [C.POINTER(OSpline), C.c_double]
        L.ospline_shift_time.argtypes = [C.POINTER(OSpline), C.c_double]
        L.ophilox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ogaussian_pair.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, c_f64p]
        L.onoise_candidate.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxNoiseSpec), C.c_int, c_f64p, C.c_int, c_f64p]
        L.orollout_spline.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), C.c_void_p, c_f64p, C.c_double,
                                      c_f64p, c_f64p, C.c_int, C.POINTER(OSpline), C.POINTER(MjpcxTrajView)]
        L.orollout_pd.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), C.c_void_p, c_f64p, C.c_double,
                                  c_f64p, C.c_int, c_f64p, c_f64p, C.c_double, C.c_double, C.POINTER(MjpcxTrajView)]
        L.orollout_batch.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, c_f64p,
                                     C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p, C.c_int, C.POINTER(OBatchOut)]
        L.odata_xfrc_applied.restype = c_f64p
        L.odata_xfrc_applied.argtypes = [C.c_void_p]
        L.orollout_batch_noisy.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, c_f64p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p, C.c_double, C.c_double,
                                           C.c_uint64, C.c_int, C.c_int, C.POINTER(OBatchOut)]
        L.otransition_fd.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), C.c_void_p, c_f64p, C.c_double, c_f64p,
                                     C.c_double, C.c_int, c_f64p, c_f64p, c_f64p, c_f64p]
        L.ocost_derivatives.argtypes = [C.POINTER(MjpcxTask), C.c_int, C.c_int, C.c_int] + [c_f64p] * 8
        L.orollout_feedback.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [c_f64p] * 6 + [C.POINTER(OBatchOut)]
        L.orollout_feedback_mt.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_double, c_f64p, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [c_f64p] * 6 + [C.c_int, C.POINTER(OBatchOut)]
        L.otransition_fd_batch.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), c_f64p, C.c_int, c_f64p, c_f64p, c_f64p,
                                           C.c_double, C.c_int, c_f64p, c_f64p, c_f64p, c_f64p, C.c_int]
        L.oriccati.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int] + [c_f64p] * 14
        L.oboxqp.argtypes = [c_f64p, c_f64p, c_i32p, c_f64p, c_f64p, C.c_int, c_f64p, c_f64p]
    return L


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Spline:
    """mjpc::spline::TimeSpline restatement."""

    def __init__(self, dim, interp=0):
        self.s = OSpline()
        lib().ospline_init(C.byref(self.s), dim, interp)
        self.dim = dim

    def __del__(self):
        try:
            lib().ospline_free(C.byref(self.s))
        except Exception:
            pass

    def size(self):
        return self.s.size

    def set_interpolation(self, interp):
        lib().ospline_set_interpolation(C.byref(self.s), interp)

    def clear(self):
        lib().ospline_clear(C.byref(self.s))

    def add_node(self, time, values=None):
        v = None if values is None else as_f64p(_f(values))
        r = lib().ospline_add_node(C.byref(self.s), float(time), v)
        if r < 0:
            raise ValueError("Adding nodes to the middle of the spline isn't supported.")
        return r

    def sample(self, time):
        out = np.zeros(max(self.dim, 1))
        lib().ospline_sample(C.byref(self.s), float(time), as_f64p(out))
        return out[:self.dim]

    def discard_before(self, time):
        return lib().ospline_discard_before(C.byref(self.s), float(time))

    def shift_time(self, t):
        lib().ospline_shift_time(C.byref(self.s), float(t))

    def copy(self):
        o = Spline(self.dim)
        lib().ospline_copy(C.byref(o.s), C.byref(self.s))
        return o

    def node_times(self):
        return np.array([self.s.times[i] for i in range(self.s.size)])

    def node_values(self):
        return np.array([self.s.values[i] for i in range(self.s.size * self.dim)]).reshape(self.s.size, self.dim)

    def set_node_values(self, idx, values):
        for k, v in enumerate(values):
            self.s.values[idx * self.dim + k] = float(v)


def norm(x, params, norm_type, grad=False, hess=False):
    x = _f(x)
    n = x.size
    p = _f(list(params) + [0.0, 0.0])
    g = np.zeros(n) if (grad or hess) else None
    H = np.zeros(n * n) if hess else None
    y = lib().onorm(as_f64p(g) if g is not None else None, as_f64p(H) if H is not None else None,
                    as_f64p(x), as_f64p(p), n, int(norm_type))
    return (y, g, None if H is None else H.reshape(n, n))


class Physics:
    """One physics arena (mjData analogue) bound to a packed model."""

    def __init__(self, packed_model):
        self.pm = packed_model
        self.m = packed_model.struct
        self.d = lib().odata_new(packed_model.ptr)
        if not self.d:
            raise NotImplementedError("model uses features outside the oracle's physics subset")

    def __del__(self):
        try:
            lib().odata_free(self.d)
        except Exception:
            pass

    def set_state(self, qpos, qvel, time=0.0, mocap=None):
        st = _f(np.concatenate([qpos, qvel]))
        mc = None if mocap is None else as_f64p(_f(mocap))
        lib().odata_set_state(self.d, as_f64p(st), float(time), mc, None)

    def set_ctrl(self, ctrl):
        lib().odata_set_ctrl(self.d, as_f64p(_f(ctrl)))

    def forward(self):
        lib().o_forward(self.d)

    def step(self):
        lib().o_step(self.d)

    def solve_pgs(self, max_sweeps=200000, tol=1e-13):
        """tests only: the constraint problem of the last forward() re-solved in its dual form by projected Gauss-Seidel
        (contact.inc o_solve_pgs) -> (qacc, efc_force, sweeps)"""
        nefc = int(self.get("nefc")[0])
        qacc, force = np.zeros(self.m.nv), np.zeros(max(nefc, 1))
        sweeps = lib().o_solve_pgs(self.d, int(max_sweeps), float(tol), qacc.ctypes.data, force.ctypes.data)
        return qacc, force[:nefc], sweeps

    def forward_task(self, packed_task):
        """mj_forward with the task residual evaluated at the sensor-callback point"""
        r = np.zeros(packed_task.struct.num_residual)
        lib().o_forward_task(self.d, packed_task.ptr, as_f64p(r))
        return r

    def warning(self):
        return lib().odata_warning(self.d)

    def get(self, name, cap=4096):
        out = np.zeros(cap)
        n = lib().odata_get(self.d, name.encode(), as_f64p(out), cap)
        if n < 0:
            raise KeyError(name)
        return out[:n].copy()


def cost_value(packed_task, residual):
    return lib().ocost_value(packed_task.ptr, as_f64p(_f(residual)))


def cost_terms(packed_task, residual, weighted=True):
    out = np.zeros(packed_task.struct.num_term)
    lib().ocost_terms(packed_task.ptr, as_f64p(_f(residual)), as_f64p(out), int(weighted))
    return out


class Trajectory:
    """mjpc::Trajectory buffers (mjpc/trajectory.h:74-86), reference layout."""

    def __init__(self, dim_state, nu, nr, ntrace, horizon):
        self.horizon = horizon
        self.states = np.zeros((horizon, dim_state))
        self.actions = np.zeros((horizon, nu))
        self.times = np.zeros(horizon)
        self.residual = np.zeros((horizon, nr))
        self.costs = np.zeros(horizon)
        self.trace = np.zeros((horizon, 3 * ntrace))
        self.total_return = 0.0
        self.failure = False

    def view(self):
        v = MjpcxTrajView()
        v.horizon = self.horizon
        v.states, v.actions, v.times = as_f64p(self.states), as_f64p(self.actions), as_f64p(self.times)
        v.residual, v.costs, v.trace = as_f64p(self.residual), as_f64p(self.costs), as_f64p(self.trace)
        return v

    def take(self, v):
        self.total_return = v.total_return
        self.failure = bool(v.failure)


def rollout_spline(pm, pt, physics, state, time, mocap, horizon, spline: Spline):
    m = pm.struct
    tr = Trajectory(m.nq + m.nv + m.na, m.nu, pt.struct.num_residual, pt.struct.num_trace, horizon)
    v = tr.view()
    mc = None if mocap is None else as_f64p(_f(mocap))
    lib().orollout_spline(pm.ptr, pt.ptr, physics.d, as_f64p(_f(state)), float(time), mc, None, horizon,
                          C.byref(spline.s), C.byref(v))
    tr.take(v)
    return tr


def rollout_pd(pm, pt, physics, state, time, mocap, horizon, pos_goal, vel_goal, P, D):
    m = pm.struct
    tr = Trajectory(m.nq + m.nv + m.na, m.nu, pt.struct.num_residual, pt.struct.num_trace, horizon)
    v = tr.view()
    lib().orollout_pd(pm.ptr, pt.ptr, physics.d, as_f64p(_f(state)), float(time), as_f64p(_f(mocap)), horizon,
                      as_f64p(_f(pos_goal)), as_f64p(_f(vel_goal)), float(P), float(D), C.byref(v))
    tr.take(v)
    return tr


def rollout_batch(pm, pt, state, time, mocap, N, H, P, interp, node_times, node_values, num_threads=1, full=True,
                  xfrc_std=0.0, xfrc_rate=1.0, seed=0, candidate_offset=0):
    """N x Trajectory::Rollout (NoisyRollout when xfrc_std > 0) over a worker pool. Returns dict of arrays (candidate-major)."""
    m = pm.struct
    ds, nu, nr, ntr = m.nq + m.nv + m.na, m.nu, pt.struct.num_residual, pt.struct.num_trace
    out = dict(total_return=np.zeros(N), failure=np.zeros(N, np.int32))
    o = OBatchOut()
    o.total_return, o.failure = as_f64p(out["total_return"]), as_i32p(out["failure"])
    if full:
        out.update(states=np.zeros((N, H, ds)), actions=np.zeros((N, H, nu)), times=np.zeros((N, H)),
                   residual=np.zeros((N, H, nr)), costs=np.zeros((N, H)), trace=np.zeros((N, H, 3 * ntr)))
        o.states, o.actions, o.times = as_f64p(out["states"]), as_f64p(out["actions"]), as_f64p(out["times"])
        o.residual, o.costs, o.trace = as_f64p(out["residual"]), as_f64p(out["costs"]), as_f64p(out["trace"])
    nt, nvv = _f(node_times), _f(node_values)
    assert nvv.size == N * P * nu
    mc = None if mocap is None else as_f64p(_f(mocap))
    lib().orollout_batch_noisy(pm.ptr, pt.ptr, as_f64p(_f(state)), float(time), mc, None, N, H, P, interp,
                               as_f64p(nt), as_f64p(nvv), float(xfrc_std), float(xfrc_rate), int(seed), int(candidate_offset),
                               int(num_threads), C.byref(o))
    return out


def noise_candidates(pm, noise_spec: MjpcxNoiseSpec, P, nominal, candidates):
    """clamp(nominal + noise) for each global candidate index in `candidates`: (len, P, nu)."""
    nu = pm.struct.nu
    nom = _f(nominal).reshape(-1)
    out = np.zeros((len(candidates), P, nu))
    for k, gi in enumerate(candidates):
        lib().onoise_candidate(pm.ptr, C.byref(noise_spec), P, as_f64p(nom), int(gi), as_f64p(out[k]))
    return out


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().ophilox4x32_10(c, k, o)
    return list(o)


def riccati(n, m, T, mu, reg_type, use_limits, A, B, cx, cu, cxx, cxu, cuu, actions, limits):
    """iLQGBackwardPass::Riccati at one regularisation value. Returns dict(ok, Vx, Vxx, K, du, dV)."""
    Vx, Vxx, K, du, dV = np.zeros(T * n), np.zeros(T * n * n), np.zeros(T * m * n), np.zeros(T * m), np.zeros(2)
    args = [_f(x).reshape(-1) for x in (A, B, cx, cu, cxx, cxu, cuu, actions, limits)]
    ok = lib().oriccati(n, m, T, float(mu), int(reg_type), int(use_limits), *[as_f64p(a) for a in args],
                        as_f64p(Vx), as_f64p(Vxx), as_f64p(K), as_f64p(du), as_f64p(dV))
    return dict(ok=bool(ok), Vx=Vx.reshape(T, n), Vxx=Vxx.reshape(T, n, n), K=K.reshape(T, m, n), du=du.reshape(T, m), dV=dV)


def boxqp(H, g, lower, upper, x0=None):
    n = len(g)
    res = np.zeros(n) if x0 is None else _f(x0).copy()
    R = np.zeros(n * (n + 7))
    idx = np.zeros(max(n, 1), np.int32)
    nfree = lib().oboxqp(as_f64p(res), as_f64p(R), as_i32p(idx), as_f64p(_f(H).reshape(-1)), as_f64p(_f(g)), n,
                         as_f64p(_f(lower)), as_f64p(_f(upper)))
    return nfree, res, idx[:max(nfree, 0)]


def transition_fd(pm, pt, states, times, actions, eps=1e-6, centered=0, mocap=None, num_threads=1):
    """ModelDerivatives::Compute: A (T,ndx,ndx), B (T,ndx,nu), C (T,nr,ndx), D (T,nr,nu); the time steps fanned over num_threads
    workers as the reference schedules them on its ThreadPool (the numbers do not depend on the thread count)."""
    m = pm.struct
    T, ndx, nu, nr = len(times), 2 * m.nv, m.nu, pt.struct.num_residual
    A, B, Cm, D = np.zeros((T, ndx, ndx)), np.zeros((T, ndx, nu)), np.zeros((T, nr, ndx)), np.zeros((T, nr, nu))
    states, actions, times = _f(states).reshape(T, -1), _f(actions).reshape(T, -1), _f(times).reshape(-1)
    mc = None if mocap is None else as_f64p(_f(mocap))
    rc = lib().otransition_fd_batch(pm.ptr, pt.ptr, mc, T, as_f64p(states), as_f64p(times), as_f64p(actions), float(eps), int(centered),
                                    as_f64p(A), as_f64p(B), as_f64p(Cm), as_f64p(D), int(num_threads))
    assert rc == 0
    return A, B, Cm, D


def cost_derivatives(pt, residual, Cm, D):
    T, nr = residual.shape
    ndx, nu = Cm.shape[2], D.shape[2]
    cx, cu = np.zeros((T, ndx)), np.zeros((T, nu))
    cxx, cxu, cuu = np.zeros((T, ndx, ndx)), np.zeros((T, ndx, nu)), np.zeros((T, nu, nu))
    for t in range(T):
        lib().ocost_derivatives(pt.ptr, T, ndx, nu, as_f64p(_f(residual[t])), as_f64p(_f(Cm[t])), as_f64p(_f(D[t])),
                                as_f64p(cx[t]), as_f64p(cu[t]), as_f64p(cxx[t]), as_f64p(cxu[t]), as_f64p(cuu[t]))
    return cx, cu, cxx, cxu, cuu


def rollout_feedback(pm, pt, state, time, mocap, H, mode, representation, use_state, times, states, actions, gains,
                     improvement, alpha, num_threads=1):
    m = pm.struct
    N, Tn = len(alpha), len(times)
    ds, nu, nr, ntr = m.nq + m.nv, m.nu, pt.struct.num_residual, pt.struct.num_trace
    out = dict(total_return=np.zeros(N), failure=np.zeros(N, np.int32), states=np.zeros((N, H, ds)), actions=np.zeros((N, H, nu)),
               times=np.zeros((N, H)), residual=np.zeros((N, H, nr)), costs=np.zeros((N, H)), trace=np.zeros((N, H, 3 * ntr)))
    o = OBatchOut()
    o.total_return, o.failure = as_f64p(out["total_return"]), as_i32p(out["failure"])
    o.states, o.actions, o.times = as_f64p(out["states"]), as_f64p(out["actions"]), as_f64p(out["times"])
    o.residual, o.costs, o.trace = as_f64p(out["residual"]), as_f64p(out["costs"]), as_f64p(out["trace"])
    mc = None if mocap is None else as_f64p(_f(mocap))
    arrs = [_f(x).reshape(-1) for x in (times, states, actions, gains, improvement, alpha)]
    rc = lib().orollout_feedback_mt(pm.ptr, pt.ptr, as_f64p(_f(state)), float(time), mc, N, H, int(mode), int(representation),
                                    int(use_state), Tn, *[as_f64p(a) for a in arrs], int(num_threads), C.byref(o))
    assert rc == 0
    return out
