"""ctypes binding of the C ABI (include/mjpcx.h) exported by libmjpcx.so.

There is no fallback: if the HIP library is missing or fails to load, importing
the binding raises. The library is built in-tree by `__graft_entry__.build()` /
`mujoco_mpc_amd.build.build_native()`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .cstructs import (MjpcxModel, MjpcxNoiseSpec, MjpcxTask, MjpcxTrajView, PackedModel, PackedTask, as_f64p,
                       as_i32p, c_f64p, c_i32p)

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmjpcx.so")

SPLINE_ZERO, SPLINE_LINEAR, SPLINE_CUBIC = 0, 1, 2
NOISE_SAMPLING, NOISE_CROSS_ENTROPY = 0, 1

# every symbol include/mjpcx.h declares
EXPORTS = [
    "mjpcx_create", "mjpcx_destroy", "mjpcx_create_error", "mjpcx_error_string", "mjpcx_last_error",
    "mjpcx_kernel_name", "mjpcx_set_state", "mjpcx_set_task_params", "mjpcx_set_residual_state", "mjpcx_rollout_splines",
    "mjpcx_rollout_noise", "mjpcx_rollout_splines_noisy", "mjpcx_kinematics", "mjpcx_sync", "mjpcx_get_returns", "mjpcx_get_return_at", "mjpcx_best", "mjpcx_topk", "mjpcx_elite_moments", "mjpcx_fetch_trajectory",
    "mjpcx_fetch_spline", "mjpcx_rollout_feedback", "mjpcx_transition_fd", "mjpcx_cost_derivatives",
    "mjpcx_backward_pass", "mjpcx_timing_reset", "mjpcx_timing_read", "mjpcx_timing_read_main", "mjpcx_quad_stats", "mjpcx_algorithmic_bytes",
    "mjpcx_device_buffer", "mjpcx_comm_unique_id", "mjpcx_comm_init", "mjpcx_comm_info", "mjpcx_exchange_best", "mjpcx_merge_topk",
    "mjpcx_elite_allreduce", "mjpcx_comm_barrier", "mjpcx_comm_destroy",
]

_LIB = None


class MjpcxError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        super().__init__(f"mjpcx error {code} ({lib().mjpcx_error_string(code).decode()}): {detail}")


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build the HIP library first "
                              "(python -c 'import __graft_entry__ as g; g.build()')")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.mjpcx_create.argtypes = [C.POINTER(MjpcxModel), C.POINTER(MjpcxTask), C.c_int, C.c_int, C.POINTER(vp)]
        L.mjpcx_destroy.argtypes = [vp]
        L.mjpcx_destroy.restype = None
        for f in ("mjpcx_create_error",):
            getattr(L, f).restype = C.c_char_p
            getattr(L, f).argtypes = []
        L.mjpcx_error_string.restype = C.c_char_p
        L.mjpcx_error_string.argtypes = [C.c_int]
        L.mjpcx_last_error.restype = C.c_char_p
        L.mjpcx_last_error.argtypes = [vp]
        L.mjpcx_kernel_name.restype = C.c_char_p
        L.mjpcx_kernel_name.argtypes = [vp]
        L.mjpcx_set_state.argtypes = [vp, c_f64p, C.c_double, c_f64p, c_f64p]
        L.mjpcx_set_task_params.argtypes = [vp, c_f64p, c_f64p, c_f64p, C.c_double]
        L.mjpcx_set_residual_state.argtypes = [vp, c_i32p, c_f64p]
        L.mjpcx_rollout_splines.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p]
        L.mjpcx_rollout_noise.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p, C.POINTER(MjpcxNoiseSpec)]
        L.mjpcx_rollout_splines_noisy.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, c_f64p, C.c_double, C.c_double,
                                                  C.c_uint64, C.c_int]
        L.mjpcx_kinematics.argtypes = [vp] + [c_f64p] * 7
        L.mjpcx_sync.argtypes = [vp]
        L.mjpcx_get_returns.argtypes = [vp, c_f64p, c_i32p]
        L.mjpcx_get_return_at.argtypes = [vp, C.c_int, C.POINTER(C.c_double), c_i32p]
        L.mjpcx_best.argtypes = [vp, C.c_int, c_i32p, C.POINTER(C.c_double), C.POINTER(C.c_double), c_f64p]
        L.mjpcx_topk.argtypes = [vp, C.c_int, c_i32p, c_f64p]
        L.mjpcx_elite_moments.argtypes = [vp, C.c_int, c_i32p, c_f64p, c_f64p, C.POINTER(C.c_double)]
        L.mjpcx_fetch_trajectory.argtypes = [vp, C.c_int, C.POINTER(MjpcxTrajView)]
        L.mjpcx_fetch_spline.argtypes = [vp, C.c_int, c_f64p]
        L.mjpcx_rollout_feedback.argtypes = [vp] + [C.c_int] * 6 + [c_f64p] * 6
        L.mjpcx_transition_fd.argtypes = [vp, C.c_int, c_f64p, c_f64p, c_f64p, C.c_double, C.c_int, c_f64p, c_f64p, c_f64p, c_f64p]
        L.mjpcx_cost_derivatives.argtypes = [vp, C.c_int] + [c_f64p] * 8
        L.mjpcx_backward_pass.argtypes = ([vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int] + [c_f64p] * 14 +
                                          [c_i32p, C.POINTER(C.c_double)])
        L.mjpcx_timing_reset.argtypes = [vp]
        L.mjpcx_timing_read.argtypes = [vp, c_f64p, C.POINTER(C.c_int64)]
        L.mjpcx_timing_read_main.argtypes = [vp, c_f64p, C.POINTER(C.c_int64)]
        L.mjpcx_quad_stats.argtypes = [vp, c_i32p]
        L.mjpcx_algorithmic_bytes.restype = C.c_int64
        L.mjpcx_algorithmic_bytes.argtypes = [vp, C.c_int, C.c_int]
        L.mjpcx_device_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
        _LIB = L
    return _LIB


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def make_noise_spec(seed=0, iteration=0, mode=NOISE_SAMPLING, candidate_offset=0, nominal_candidate=0,
                    explore_count=0, std0=0.1, std1=0.0, param_variance=None):
    ns = MjpcxNoiseSpec()
    ns.seed, ns.iteration, ns.mode = int(seed), int(iteration), int(mode)
    ns.candidate_offset, ns.nominal_candidate, ns.explore_count = int(candidate_offset), int(nominal_candidate), int(explore_count)
    ns.std0, ns.std1 = float(std0), float(std1)
    keep = None
    if param_variance is not None:
        keep = _f(param_variance).reshape(-1)
        ns.param_variance = as_f64p(keep)
    ns._keep = keep
    return ns


class Trajectory:
    """mjpc::Trajectory buffers in the reference layout (mjpc/trajectory.h:74-86)."""

    def __init__(self, dim_state, nu, nr, ntrace, horizon):
        self.horizon = horizon
        self.dim_state, self.dim_action, self.dim_residual, self.dim_trace = dim_state, nu, nr, 3 * ntrace
        self.states = np.zeros((horizon, dim_state))
        self.actions = np.zeros((horizon, nu))
        self.times = np.zeros(horizon)
        self.residual = np.zeros((horizon, nr))
        self.costs = np.zeros(horizon)
        self.trace = np.zeros((horizon, 3 * ntrace))
        self.total_return = 0.0
        self.failure = False


class Context:
    """One (model, task, device) rollout context = `mjpcx_ctx`."""

    def __init__(self, packed_model: PackedModel, packed_task: PackedTask, device=0, precision=64):
        self._pm, self._pt = packed_model, packed_task
        self.handle = C.c_void_p()
        rc = lib().mjpcx_create(packed_model.ptr, packed_task.ptr, int(device), int(precision), C.byref(self.handle))
        if rc != 0:
            raise MjpcxError(rc, lib().mjpcx_create_error().decode())
        self.create_warning = lib().mjpcx_create_error().decode()  # "" or what of the model the device does not reproduce
        m, t = packed_model.struct, packed_task.struct
        self.nq, self.nv, self.nu, self.na = m.nq, m.nv, m.nu, m.na
        self.dim_state = m.nq + m.nv + m.na
        self.num_residual, self.num_trace = t.num_residual, t.num_trace
        self.precision = precision
        self.device = device
        self.N = self.H = self.P = 0

    def close(self):
        if self.handle:
            lib().mjpcx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise MjpcxError(rc, lib().mjpcx_last_error(self.handle).decode())

    @property
    def kernel_name(self):
        return lib().mjpcx_kernel_name(self.handle).decode()

    def set_state(self, state, time=0.0, mocap=None, userdata=None):
        st = _f(state)
        assert st.size == self.dim_state
        mc = None if mocap is None else as_f64p(_f(mocap))
        ud = None if userdata is None else as_f64p(_f(userdata))
        self._chk(lib().mjpcx_set_state(self.handle, as_f64p(st), float(time), mc, ud))

    def set_task_params(self, weight=None, norm_parameter=None, parameters=None, risk=0.0):
        w = None if weight is None else as_f64p(_f(weight))
        n = None if norm_parameter is None else as_f64p(_f(norm_parameter))
        p = None if parameters is None else as_f64p(_f(parameters))
        self._chk(lib().mjpcx_set_task_params(self.handle, w, n, p, float(risk)))

    def set_residual_state(self, residual_int=None, residual_real=None):
        ri = None if residual_int is None else np.ascontiguousarray(residual_int, dtype=np.int32)
        rr = None if residual_real is None else _f(residual_real)
        self._chk(lib().mjpcx_set_residual_state(self.handle, None if ri is None else ri.ctypes.data_as(c_i32p),
                                                 None if rr is None else as_f64p(rr)))

    def rollout_splines(self, horizon, interp, node_times, node_values):
        nt = _f(node_times)
        nv = _f(node_values)
        P = nt.size
        N = nv.size // (P * self.nu)
        assert nv.size == N * P * self.nu
        self._chk(lib().mjpcx_rollout_splines(self.handle, N, int(horizon), P, int(interp), as_f64p(nt), as_f64p(nv)))
        self.N, self.H, self.P = N, int(horizon), P

    def rollout_splines_noisy(self, horizon, interp, node_times, node_values, xfrc_std, xfrc_rate, seed=0, candidate_offset=0):
        """Trajectory::NoisyRollout for every candidate spline (Ornstein-Uhlenbeck xfrc_applied noise)."""
        nt = _f(node_times)
        nv = _f(node_values)
        P = nt.size
        N = nv.size // (P * self.nu)
        assert nv.size == N * P * self.nu
        self._chk(lib().mjpcx_rollout_splines_noisy(self.handle, N, int(horizon), P, int(interp), as_f64p(nt), as_f64p(nv),
                                                    float(xfrc_std), float(xfrc_rate), int(seed), int(candidate_offset)))
        self.N, self.H, self.P = N, int(horizon), P

    def kinematics(self, nbody, nsite):
        """mjData kinematics of the state given to set_state: dict of xpos, xquat, xmat, xipos, site_xpos, subtree_com, subtree_linvel."""
        out = dict(xpos=np.zeros((nbody, 3)), xquat=np.zeros((nbody, 4)), xmat=np.zeros((nbody, 9)), xipos=np.zeros((nbody, 3)),
                   site_xpos=np.zeros((nsite, 3)), subtree_com=np.zeros((nbody, 3)), subtree_linvel=np.zeros((nbody, 3)))
        self._chk(lib().mjpcx_kinematics(self.handle, *[as_f64p(out[k]) for k in ("xpos", "xquat", "xmat", "xipos", "site_xpos", "subtree_com",
                                                                                   "subtree_linvel")]))
        return out

    def rollout_noise(self, num_candidates, horizon, interp, node_times, nominal, noise_spec):
        nt, nom = _f(node_times), _f(nominal)
        P = nt.size
        assert nom.size == P * self.nu
        self._chk(lib().mjpcx_rollout_noise(self.handle, int(num_candidates), int(horizon), P, int(interp),
                                            as_f64p(nt), as_f64p(nom), C.byref(noise_spec)))
        self.N, self.H, self.P = int(num_candidates), int(horizon), P

    def sync(self):
        self._chk(lib().mjpcx_sync(self.handle))

    def returns(self):
        ret = np.zeros(self.N)
        fl = np.zeros(self.N, np.int32)
        self._chk(lib().mjpcx_get_returns(self.handle, as_f64p(ret), as_i32p(fl)))
        # non-zero = failed (Trajectory::failure); the wavefront kernels add diagnostics above the low byte
        # (warning bits << 8 | failing step << 16), kept for tools in failure_raw
        self.failure_raw = fl.copy()
        return ret, (fl != 0).astype(np.int32)

    def return_of(self, candidate):
        r = C.c_double()
        self._chk(lib().mjpcx_get_return_at(self.handle, int(candidate), C.byref(r), None))
        return r.value

    def best(self, ref_candidate=0, with_spline=True):
        """argmin + winner spline + reference candidate's return in one launch / one sync."""
        idx = np.zeros(1, np.int32)
        br, rr = C.c_double(), C.c_double()
        sp = np.zeros((self.P, self.nu)) if with_spline else None
        self._chk(lib().mjpcx_best(self.handle, int(ref_candidate), as_i32p(idx), C.byref(br), C.byref(rr),
                                   as_f64p(sp) if with_spline else None))
        return int(idx[0]), br.value, rr.value, sp

    def topk(self, k):
        idx = np.zeros(k, np.int32)
        ret = np.zeros(k)
        self._chk(lib().mjpcx_topk(self.handle, int(k), as_i32p(idx), as_f64p(ret)))
        return idx, ret

    def elite_moments(self, candidates, mean=None):
        """(sum over the listed local candidates of p, or of (p - mean)^2) per spline parameter, and sum of returns."""
        cand = np.ascontiguousarray(candidates, dtype=np.int32).reshape(-1)
        out = np.zeros(self.P * self.nu)
        sr = C.c_double()
        m = None if mean is None else as_f64p(_f(mean).reshape(-1))
        self._chk(lib().mjpcx_elite_moments(self.handle, cand.size, as_i32p(cand) if cand.size else as_i32p(np.zeros(1, np.int32)),
                                            m, as_f64p(out), C.byref(sr)))
        return out.reshape(self.P, self.nu), sr.value

    def fetch_trajectory(self, candidate) -> Trajectory:
        tr = Trajectory(self.dim_state, self.nu, self.num_residual, self.num_trace, self.H)
        v = MjpcxTrajView()
        v.horizon = self.H
        v.states, v.actions, v.times = as_f64p(tr.states), as_f64p(tr.actions), as_f64p(tr.times)
        v.residual, v.costs, v.trace = as_f64p(tr.residual), as_f64p(tr.costs), as_f64p(tr.trace)
        self._chk(lib().mjpcx_fetch_trajectory(self.handle, int(candidate), C.byref(v)))
        tr.total_return, tr.failure = v.total_return, bool(v.failure)
        return tr

    def fetch_spline(self, candidate):
        out = np.zeros((self.P, self.nu))
        self._chk(lib().mjpcx_fetch_spline(self.handle, int(candidate), as_f64p(out)))
        return out

    # ---- iLQG
    def rollout_feedback(self, horizon, mode, representation, use_state, times, states, actions, gains, improvement, alpha):
        arrs = [_f(x).reshape(-1) for x in (times, states, actions, gains, improvement, alpha)]
        N, Tn = arrs[5].size, arrs[0].size
        self._chk(lib().mjpcx_rollout_feedback(self.handle, N, int(horizon), int(mode), int(representation), int(use_state),
                                               Tn, *[as_f64p(a) for a in arrs]))
        self.N, self.H, self.P = N, int(horizon), 0

    def transition_fd(self, times, states, actions, eps=1e-6, centered=0):
        T, ndx, nu, nr = len(times), 2 * self.nv, self.nu, self.num_residual
        A, B, Cm, D = np.zeros((T, ndx, ndx)), np.zeros((T, ndx, nu)), np.zeros((T, nr, ndx)), np.zeros((T, nr, nu))
        self._chk(lib().mjpcx_transition_fd(self.handle, T, as_f64p(_f(times)), as_f64p(_f(states).reshape(-1)),
                                            as_f64p(_f(actions).reshape(-1)), float(eps), int(centered), as_f64p(A),
                                            as_f64p(B), as_f64p(Cm), as_f64p(D)))
        return A, B, Cm, D

    def cost_derivatives(self, residual, Cm, D):
        T, ndx, nu = residual.shape[0], 2 * self.nv, self.nu
        cx, cu = np.zeros((T, ndx)), np.zeros((T, nu))
        cxx, cxu, cuu = np.zeros((T, ndx, ndx)), np.zeros((T, ndx, nu)), np.zeros((T, nu, nu))
        self._chk(lib().mjpcx_cost_derivatives(self.handle, T, as_f64p(_f(residual).reshape(-1)), as_f64p(_f(Cm).reshape(-1)),
                                               as_f64p(_f(D).reshape(-1)), as_f64p(cx), as_f64p(cu), as_f64p(cxx),
                                               as_f64p(cxu), as_f64p(cuu)))
        return cx, cu, cxx, cxu, cuu

    def backward_pass(self, mu, reg_type, use_limits, A, B, cx, cu, cxx, cxu, cuu, actions, limits):
        T, n, m = A.shape[0], A.shape[1], B.shape[2]
        Vx, Vxx, K, du, dV = np.zeros((T, n)), np.zeros((T, n, n)), np.zeros((T, m, n)), np.zeros((T, m)), np.zeros(2)
        st = np.zeros(1, np.int32)
        ms = C.c_double()
        args = [as_f64p(_f(x).reshape(-1)) for x in (A, B, cx, cu, cxx, cxu, cuu, actions, limits)]
        self._chk(lib().mjpcx_backward_pass(self.handle, n, m, T, float(mu), int(reg_type), int(use_limits), *args,
                                            as_f64p(Vx), as_f64p(Vxx), as_f64p(K), as_f64p(du), as_f64p(dV), as_i32p(st),
                                            C.byref(ms)))
        return dict(ok=bool(st[0]), Vx=Vx, Vxx=Vxx, K=K, du=du, dV=dV, kernel_ms=ms.value)

    def timing_reset(self):
        self._chk(lib().mjpcx_timing_reset(self.handle))

    def timing_read(self):
        ms = C.c_double()
        n = C.c_int64()
        self._chk(lib().mjpcx_timing_read(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timing_read_main(self):
        """HIP-event time of the rollouts' first (dominant) kernel alone; call before timing_read"""
        ms = C.c_double()
        n = C.c_int64()
        self._chk(lib().mjpcx_timing_read_main(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def quad_stats(self):
        """rollout_quad_kernel: candidates of the last rollout handed to the wavefront-per-candidate kernel, total and by reason"""
        h = np.zeros(8, np.int32)
        self._chk(lib().mjpcx_quad_stats(self.handle, as_i32p(h)))
        return dict(handed_on=int(h[0]), contact_list_full=int(h[1]), leg_leg_contact=int(h[2]), indefinite_hessian=int(h[3]), non_finite=int(h[4]),
                    both_limits=int(h[5]), trunk_leg_contact=int(h[6]), out_of_proof_range=int(h[7]))

    def algorithmic_bytes(self, horizon, num_nodes):
        return lib().mjpcx_algorithmic_bytes(self.handle, int(horizon), int(num_nodes))

    def device_buffer(self, which):
        p = C.c_void_p()
        n = C.c_size_t()
        self._chk(lib().mjpcx_device_buffer(self.handle, int(which), C.byref(p), C.byref(n)))
        return p.value, n.value
