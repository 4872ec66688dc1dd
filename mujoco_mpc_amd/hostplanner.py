"""ctypes view of the C++ planners `mjpc::GpuSamplingPlanner`, `mjpc::GpuCrossEntropyPlanner` and `mjpc::GpuILQGPlanner`
(mujoco_mpc_amd/host, planner_c_api.cc).

The planner logic (nominal resampling, policy bookkeeping, C-ABI calls) is the C++ host layer; Python
only drives it, and -- for several ranks -- lends it a transport for the per-step candidate exchange
(`torch.distributed`: RCCL over xGMI with backend "nccl", gloo in CPU-side tests)."""
from __future__ import annotations

import ctypes as C
import os
import tempfile

import numpy as np

from . import capi, mjcf
from .build import build_host
from .cstructs import as_f64p, c_f64p

_LIB = None
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)
MERGE_TOPK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double))
SUM_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


def lib():
    global _LIB
    if _LIB is None:
        capi.lib()  # libmjpcx.so first (the host library links it)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "build", "libmjpc_host.so")
        if not os.path.exists(path):
            build_host()
        L = C.CDLL(path)
        vp = C.c_void_p
        L.mjpc_planner_create.restype = vp
        L.mjpc_planner_create.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_ulonglong, C.c_int]
        L.mjpc_planner_create_kind.restype = vp
        L.mjpc_planner_create_kind.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_ulonglong, C.c_int]
        L.mjpc_planner_set_sharding_ce.argtypes = [vp, C.c_int, C.c_int, MERGE_TOPK_FN, SUM_FN, vp]
        L.mjpc_planner_nominal.argtypes = [vp, C.c_int]
        L.mjpc_planner_action_state.argtypes = [vp, c_f64p, C.c_double, C.c_int, c_f64p]
        L.mjpc_planner_num_parameters.argtypes = [vp]
        L.mjpc_planner_ce_variance.argtypes = [vp, c_f64p, C.c_int]
        L.mjpc_planner_ce_elites.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
        L.mjpc_planner_ce_set.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]
        L.mjpc_planner_ilqg_info.argtypes = [vp, c_f64p]
        L.mjpc_planner_ilqg_set.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.mjpc_planner_ilqg_policy.argtypes = [vp, C.c_int, c_f64p, c_f64p, c_f64p, c_f64p]
        L.mjpc_planner_best_trajectory.argtypes = [vp, c_f64p, c_f64p, c_f64p, c_f64p, C.POINTER(C.c_double)]
        L.mjpc_planner_task_transition.argtypes = [vp, C.c_double, C.c_int]
        L.mjpc_planner_task_transition_state.argtypes = [vp, C.c_double, C.c_int, c_f64p, c_f64p, c_f64p]
        L.mjpc_host_gaussian_pair.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, c_f64p]
        L.mjpc_host_gaussian_pair.restype = None
        L.mjpc_planner_robust_config.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double]
        L.mjpc_planner_sample_gradient_config.argtypes = [vp, C.c_int, C.c_double]
        L.mjpc_planner_sample_gradient_result.argtypes = [vp, C.POINTER(C.c_int), c_f64p, C.c_int, c_f64p, C.c_int]
        L.mjpc_planner_robust_result.argtypes = [vp, C.POINTER(C.c_int), c_f64p, C.c_int]
        L.mjpc_planner_task_set_parameter.argtypes = [vp, C.c_int, C.c_double]
        L.mjpc_planner_destroy.argtypes = [vp]
        L.mjpc_planner_last_error.restype = C.c_char_p
        L.mjpc_planner_last_error.argtypes = [vp]
        L.mjpc_planner_set_sharding.argtypes = [vp, C.c_int, C.c_int, EXCHANGE_FN, vp]
        L.mjpc_planner_set_sharding_ranked.argtypes = [vp, C.c_int, C.c_int, MERGE_TOPK_FN, SUM_FN, vp]
        L.mjpc_comm_unique_id.argtypes = [C.c_void_p]
        L.mjpc_planner_comm_init.argtypes = [vp, C.c_void_p, C.c_int, C.c_int]
        L.mjpc_planner_comm_barrier.argtypes = [vp]
        L.mjpc_planner_reset.argtypes = [vp, C.c_int]
        L.mjpc_planner_set_state.argtypes = [vp, c_f64p, c_f64p, c_f64p, c_f64p, C.c_double]
        L.mjpc_planner_optimize.argtypes = [vp, C.c_int]
        L.mjpc_planner_action.argtypes = [vp, C.c_double, C.c_int, c_f64p]
        L.mjpc_planner_num_spline_points.argtypes = [vp]
        L.mjpc_planner_winner.argtypes = [vp]
        L.mjpc_planner_improvement.restype = C.c_double
        L.mjpc_planner_improvement.argtypes = [vp]
        L.mjpc_planner_best_score.restype = C.c_double
        L.mjpc_planner_best_score.argtypes = [vp]
        L.mjpc_planner_policy.argtypes = [vp, c_f64p, c_f64p, C.c_int]
        L.mjpc_planner_ctx.restype = vp
        L.mjpc_planner_ctx.argtypes = [vp]
        _LIB = L
    return _LIB


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 calls this and ships the bytes to the other ranks)"""
    buf = C.create_string_buffer(128)
    rc = lib().mjpc_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError(f"mjpcx_comm_unique_id failed ({rc}): " + capi.lib().mjpcx_create_error().decode())
    return buf.raw


def host_gaussian_pair(seed, cand, pair, iteration):
    """the C++ host's normal generator (must equal the device / oracle stream)"""
    z = np.zeros(2)
    lib().mjpc_host_gaussian_pair(int(seed), int(cand), int(pair), int(iteration), as_f64p(z))
    return z


class HostPlanner:
    def __init__(self, task, device=0, precision=64, seed=0, num_trajectory=0, group=None, kind="sampling", native_comm=None):
        """group: a RankGroup (torch.distributed) lending the transport of the per-step exchange -- the CPU-side gloo tests;
        native_comm = (unique_id bytes, rank, world): RCCL inside libmjpcx.so instead (mjpcx_comm_init), no Python on the path"""
        self.task = task
        self.kind = kind
        self.nu = task.model.nu
        self._blob = tempfile.NamedTemporaryFile(suffix=".mjpx", delete=False).name
        mjcf.save_blob(task.model, self._blob)
        self.h = lib().mjpc_planner_create_kind(kind.encode(), self._blob.encode(), task.name.encode(), device, precision, seed,
                                                num_trajectory)
        if not self.h:
            raise RuntimeError(lib().mjpc_planner_last_error(None).decode())
        self.group = group
        self._cb = None
        if native_comm is not None:
            uid, rank, world = native_comm
            buf = C.create_string_buffer(bytes(uid), 128)
            self._chk(lib().mjpc_planner_comm_init(self.h, buf, int(rank), int(world)))
            self.group = group = None
        if group is not None and group.world > 1 and kind in ("cross_entropy", "robust"):
            def merge(user, k, index, ret):
                try:
                    idx = np.ctypeslib.as_array(index, (k,))
                    r = np.ctypeslib.as_array(ret, (k,))
                    have = idx >= 0
                    gi, gr = group.merge_topk(idx[have].copy(), r[have].copy(), k)
                    idx[:] = -1
                    r[:] = 1.0e300
                    idx[:len(gi)] = gi
                    r[:len(gr)] = gr
                    return 0
                except Exception as e:
                    print("top-k exchange failed:", e, flush=True)
                    return 1

            def total(user, values, n):
                try:
                    v = np.ctypeslib.as_array(values, (n,))
                    v[:] = group.sum_array(v.copy())
                    return 0
                except Exception as e:
                    print("sum exchange failed:", e, flush=True)
                    return 1
            self._cb = (MERGE_TOPK_FN(merge), SUM_FN(total))
            # (robust: the delegate ranks its share, the k best of all ranks are merged; the perturbed rollouts are sharded the same way)
            setter = lib().mjpc_planner_set_sharding_ce if kind == "cross_entropy" else lib().mjpc_planner_set_sharding_ranked
            self._chk(setter(self.h, group.rank, group.world, self._cb[0], self._cb[1], None))
        elif group is not None and group.world > 1:
            if kind != "sampling":
                raise ValueError(f"the {kind} planner is not sharded (replicas only)")

            def exchange(user, record, spline, n):
                try:
                    rec = np.ctypeslib.as_array(record, (3,))
                    sp = np.ctypeslib.as_array(spline, (n,))
                    idx, best, nominal, vals = group.exchange_best(int(rec[1]), float(rec[0]), float(rec[2]), sp.copy())
                    rec[0], rec[1], rec[2] = best, float(idx), nominal
                    sp[:] = np.asarray(vals).reshape(-1)
                    return 0
                except Exception as e:  # never let an exception cross the C boundary
                    print("exchange failed:", e, flush=True)
                    return 1
            self._cb = EXCHANGE_FN(exchange)
            self._chk(lib().mjpc_planner_set_sharding(self.h, group.rank, group.world, self._cb, None))

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(lib().mjpc_planner_last_error(self.h).decode())

    def close(self):
        if self.h:
            lib().mjpc_planner_destroy(self.h)
            self.h = None
            try:
                os.unlink(self._blob)
            except OSError:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def comm_info(self):
        """(rank, world) of the library's own RCCL communicator (mjpcx_comm_info); (0, 1) before mjpcx_comm_init"""
        ctx = lib().mjpc_planner_ctx(self.h)
        r, w = C.c_int(0), C.c_int(1)
        if ctx:
            L = capi.lib()
            L.mjpcx_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
            L.mjpcx_comm_info(ctx, C.byref(r), C.byref(w))
        return r.value, w.value

    def comm_barrier(self):
        self._chk(lib().mjpc_planner_comm_barrier(self.h))

    def reset(self, horizon):
        self._chk(lib().mjpc_planner_reset(self.h, horizon))

    def set_state(self, qpos, qvel, time=0.0, mocap_pos=None, mocap_quat=None):
        q, v = np.ascontiguousarray(qpos, float), np.ascontiguousarray(qvel, float)
        mp = None if mocap_pos is None else as_f64p(np.ascontiguousarray(mocap_pos, float))
        mq = None if mocap_quat is None else as_f64p(np.ascontiguousarray(mocap_quat, float))
        self._chk(lib().mjpc_planner_set_state(self.h, as_f64p(q), as_f64p(v), mp, mq, float(time)))

    def optimize_policy(self, horizon):
        self._chk(lib().mjpc_planner_optimize(self.h, horizon))

    def task_transition(self, time, mode=-1):
        self._chk(lib().mjpc_planner_task_transition(self.h, float(time), int(mode)))

    def sample_gradient_config(self, num_gradient=-1, gradient_filter=-1.0):
        self._chk(lib().mjpc_planner_sample_gradient_config(self.h, int(num_gradient), float(gradient_filter)))

    def sample_gradient_result(self, num_parameters, num_trajectory):
        wt = C.c_int(-1)
        g, r = np.zeros(num_parameters), np.zeros(num_trajectory)
        self._chk(lib().mjpc_planner_sample_gradient_result(self.h, C.byref(wt), as_f64p(g), num_parameters, as_f64p(r), num_trajectory))
        return wt.value, g, r

    def robust_config(self, ncandidates=0, nrepetitions=0, xfrc_std=-1.0, xfrc_rate=0.0):
        self._chk(lib().mjpc_planner_robust_config(self.h, int(ncandidates), int(nrepetitions), float(xfrc_std), float(xfrc_rate)))

    def robust_result(self, capacity=64):
        best = C.c_int(-1)
        scores = np.zeros(capacity)
        self._chk(lib().mjpc_planner_robust_result(self.h, C.byref(best), as_f64p(scores), capacity))
        return best.value, scores

    def task_transition_state(self, time, mode, qpos, qvel, mocap_pos):
        """Task::Transition for tasks that edit the simulation state (humanoid::Tracking): the arrays are updated in place."""
        for a in (qpos, qvel, mocap_pos):
            assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
        self._chk(lib().mjpc_planner_task_transition_state(self.h, float(time), int(mode), as_f64p(qpos), as_f64p(qvel), as_f64p(mocap_pos)))

    def task_set_parameter(self, index, value):
        self._chk(lib().mjpc_planner_task_set_parameter(self.h, int(index), float(value)))

    def nominal_trajectory(self, horizon):
        self._chk(lib().mjpc_planner_nominal(self.h, horizon))

    def action(self, time, use_previous=False, state=None):
        a = np.zeros(self.nu)
        if state is None:
            self._chk(lib().mjpc_planner_action(self.h, float(time), int(use_previous), as_f64p(a)))
        else:
            st = np.ascontiguousarray(state, float)
            self._chk(lib().mjpc_planner_action_state(self.h, as_f64p(st), float(time), int(use_previous), as_f64p(a)))
        return a

    # ---- cross-entropy
    def ce_set(self, n_elite=-1, std_initial=-1.0, std_min=-1.0, explore_fraction=-1.0, interpolation=-1):
        assert lib().mjpc_planner_ce_set(self.h, n_elite, std_initial, std_min, explore_fraction, interpolation) == 0

    def ce_variance(self, n):
        v = np.zeros(n)
        assert lib().mjpc_planner_ce_variance(self.h, as_f64p(v), n) == 0
        return v

    def ce_elites(self, cap=65536):
        idx = (C.c_int * cap)()
        n = lib().mjpc_planner_ce_elites(self.h, idx, cap)
        return np.array(idx[:n], dtype=np.int64)

    # ---- iLQG
    def ilqg_set(self, regularization_type=-1, action_limits=-1, fd_mode=-1, derivative_skip=-1, representation=-1):
        assert lib().mjpc_planner_ilqg_set(self.h, regularization_type, action_limits, fd_mode, derivative_skip, representation) == 0

    def ilqg_info(self):
        v = np.zeros(10)
        assert lib().mjpc_planner_ilqg_info(self.h, as_f64p(v)) == 0
        keys = ("regularization", "dV0", "dV1", "action_step", "feedback_scaling", "improvement", "expected", "surprise",
                "winner", "total_return")
        return dict(zip(keys, v))

    def ilqg_policy(self, T):
        m = self.task.model
        ds, ndx = m.nq + m.nv + m.na, 2 * m.nv + m.na
        t, x, u, K = np.zeros(T), np.zeros((T, ds)), np.zeros((T, self.nu)), np.zeros((T, self.nu, ndx))
        lib().mjpc_planner_ilqg_policy(self.h, T, as_f64p(t), as_f64p(x), as_f64p(u), as_f64p(K))
        return t, x, u, K

    def best_trajectory(self, cap=512):
        m = self.task.model
        ds = m.nq + m.nv + m.na
        x, u, t, c = np.zeros((cap, ds)), np.zeros((cap, self.nu)), np.zeros(cap), np.zeros(cap)
        ret = C.c_double()
        T = lib().mjpc_planner_best_trajectory(self.h, as_f64p(x), as_f64p(u), as_f64p(t), as_f64p(c), C.byref(ret))
        if T < 0:
            return None
        return dict(states=x[:T], actions=u[:T], times=t[:T], costs=c[:T], total_return=ret.value)

    @property
    def num_spline_points(self):
        return lib().mjpc_planner_num_spline_points(self.h)

    @property
    def winner(self):
        return lib().mjpc_planner_winner(self.h)

    @property
    def improvement(self):
        return lib().mjpc_planner_improvement(self.h)

    @property
    def best_score(self):
        return lib().mjpc_planner_best_score(self.h)

    def policy(self):
        P = max(self.num_spline_points, 1)
        t, v = np.zeros(P), np.zeros((P, self.nu))
        n = lib().mjpc_planner_policy(self.h, as_f64p(t), as_f64p(v), P)
        return t[:n], v[:n]

    # ---- measurement hooks on the underlying mjpcx context
    def _ctx(self):
        return C.c_void_p(lib().mjpc_planner_ctx(self.h))

    def sync(self):
        capi.lib().mjpcx_sync(self._ctx())

    def timing_reset(self):
        capi.lib().mjpcx_timing_reset(self._ctx())

    def timing_read(self):
        ms, n = C.c_double(), C.c_int64()
        capi.lib().mjpcx_timing_read(self._ctx(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def timing_read_main(self):
        """HIP-event time of the rollouts' first (dominant) kernel alone (mjpcx_timing_read_main); call before timing_read"""
        ms, n = C.c_double(), C.c_int64()
        capi.lib().mjpcx_timing_read_main(self._ctx(), C.byref(ms), C.byref(n))
        return ms.value, n.value

    def quad_stats(self):
        """rollout_quad_kernel: candidates of the last rollout handed to the wavefront-per-candidate kernel ([0] total, then by reason)"""
        import numpy as np
        hh = np.zeros(8, np.int32)
        capi.lib().mjpcx_quad_stats(self._ctx(), capi.as_i32p(hh))
        return [int(x) for x in hh]

    def algorithmic_bytes(self, horizon, num_nodes):
        return capi.lib().mjpcx_algorithmic_bytes(self._ctx(), horizon, num_nodes)

    @property
    def kernel_name(self):
        return capi.lib().mjpcx_kernel_name(self._ctx()).decode()
