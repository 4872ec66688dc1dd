"""ctypes view of the C++ `mjpc::GpuSamplingPlanner` (mujoco_mpc_amd/host, planner_c_api.cc).

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
        L.mjpc_planner_destroy.argtypes = [vp]
        L.mjpc_planner_last_error.restype = C.c_char_p
        L.mjpc_planner_last_error.argtypes = [vp]
        L.mjpc_planner_set_sharding.argtypes = [vp, C.c_int, C.c_int, EXCHANGE_FN, vp]
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


class HostPlanner:
    def __init__(self, task, device=0, precision=64, seed=0, num_trajectory=0, group=None):
        self.task = task
        self.nu = task.model.nu
        self._blob = tempfile.NamedTemporaryFile(suffix=".mjpx", delete=False).name
        mjcf.save_blob(task.model, self._blob)
        self.h = lib().mjpc_planner_create(self._blob.encode(), task.name.encode(), device, precision, seed, num_trajectory)
        if not self.h:
            raise RuntimeError(lib().mjpc_planner_last_error(None).decode())
        self.group = group
        self._cb = None
        if group is not None and group.world > 1:
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

    def reset(self, horizon):
        self._chk(lib().mjpc_planner_reset(self.h, horizon))

    def set_state(self, qpos, qvel, time=0.0, mocap_pos=None, mocap_quat=None):
        q, v = np.ascontiguousarray(qpos, float), np.ascontiguousarray(qvel, float)
        mp = None if mocap_pos is None else as_f64p(np.ascontiguousarray(mocap_pos, float))
        mq = None if mocap_quat is None else as_f64p(np.ascontiguousarray(mocap_quat, float))
        self._chk(lib().mjpc_planner_set_state(self.h, as_f64p(q), as_f64p(v), mp, mq, float(time)))

    def optimize_policy(self, horizon):
        self._chk(lib().mjpc_planner_optimize(self.h, horizon))

    def action(self, time, use_previous=False):
        a = np.zeros(self.nu)
        self._chk(lib().mjpc_planner_action(self.h, float(time), int(use_previous), as_f64p(a)))
        return a

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

    def algorithmic_bytes(self, horizon, num_nodes):
        return capi.lib().mjpcx_algorithmic_bytes(self._ctx(), horizon, num_nodes)

    @property
    def kernel_name(self):
        return capi.lib().mjpcx_kernel_name(self._ctx()).decode()
