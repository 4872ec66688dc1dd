"""Test-only stand-in for capi.Context backed by the CPU oracle, so the planner's host logic and the
multi-rank exchange can be exercised without a GPU (gloo, world_size 2). Never used by the product."""
import numpy as np

from mujoco_mpc_amd import capi
from oracle import pyoracle


class OracleContext:
    def __init__(self, task, threads=2, differentiable=False):
        self.task = task
        self.pm, self.pt = task.packed_model(differentiable=differentiable), task.packed()
        self.nu = self.pm.struct.nu
        self.threads = threads
        self.N = self.H = self.P = 0

    def set_state(self, state, time=0.0, mocap=None, userdata=None):
        self.state, self.time = np.array(state, float), float(time)
        self.mocap = None if mocap is None or len(mocap) == 0 else np.array(mocap, float)

    def _run(self, N, H, interp, times, nodes):
        self.N, self.H, self.P = N, H, len(times)
        self.nodes = np.array(nodes, float).reshape(N, self.P, self.nu)
        self.out = pyoracle.rollout_batch(self.pm, self.pt, self.state, self.time, self.mocap, N, H, self.P, interp,
                                          times, self.nodes, num_threads=self.threads)

    def rollout_splines(self, horizon, interp, node_times, node_values):
        nv = np.asarray(node_values, float)
        self._run(nv.size // (len(node_times) * self.nu), horizon, interp, np.asarray(node_times, float), nv)

    def rollout_noise(self, n, horizon, interp, node_times, nominal, ns):
        P = len(node_times)
        cands = range(ns.candidate_offset, ns.candidate_offset + n)
        self._run(n, horizon, interp, np.asarray(node_times, float), pyoracle.noise_candidates(self.pm, ns, P, nominal, cands))

    def returns(self):
        return self.out["total_return"].copy(), self.out["failure"].copy()

    def return_of(self, i):
        return float(self.out["total_return"][i])

    def topk(self, k):
        r = self.out["total_return"]
        order = np.lexsort((np.arange(self.N), r))[:k]
        return order.astype(np.int32), r[order]

    def best(self, ref_candidate=0, with_spline=True):
        idx, ret = self.topk(1)
        ref = float(self.out["total_return"][ref_candidate]) if ref_candidate >= 0 else float("nan")
        return int(idx[0]), float(ret[0]), ref, self.nodes[idx[0]].copy()

    def elite_moments(self, candidates, mean=None):
        c = np.asarray(candidates, int)
        p = self.nodes[c]
        out = p.sum(axis=0) if mean is None else ((p - np.asarray(mean).reshape(1, self.P, self.nu)) ** 2).sum(axis=0)
        return out.reshape(self.P, self.nu), float(self.out["total_return"][c].sum())

    def fetch_spline(self, i):
        return self.nodes[i].copy()

    def fetch_trajectory(self, i):
        tr = capi.Trajectory(self.state.size, self.nu, self.pt.struct.num_residual, self.pt.struct.num_trace, self.H)
        for name in ("states", "actions", "times", "residual", "costs", "trace"):
            getattr(tr, name)[...] = self.out[name][i]
        tr.total_return, tr.failure = float(self.out["total_return"][i]), bool(self.out["failure"][i])
        return tr

    # ---- iLQG
    def rollout_feedback(self, horizon, mode, representation, use_state, times, states, actions, gains, improvement, alpha):
        self.N, self.H, self.P = len(alpha), horizon, 0
        self.out = pyoracle.rollout_feedback(self.pm, self.pt, self.state, self.time, self.mocap, horizon, mode, representation,
                                             use_state, times, states, actions, gains, improvement, alpha, num_threads=self.threads)

    def transition_fd(self, times, states, actions, eps=1e-6, centered=0):
        return pyoracle.transition_fd(self.pm, self.pt, states, times, actions, eps, centered, mocap=self.mocap, num_threads=self.threads)

    def cost_derivatives(self, residual, Cm, D):
        return pyoracle.cost_derivatives(self.pt, np.asarray(residual), np.asarray(Cm), np.asarray(D))

    def backward_pass(self, mu, reg_type, use_limits, A, B, cx, cu, cxx, cxu, cuu, actions, limits):
        T, n, m = A.shape[0], A.shape[1], B.shape[2]
        return pyoracle.riccati(n, m, T, mu, reg_type, use_limits, A, B, cx, cu, cxx, cxu, cuu, actions, limits)

    def sync(self):
        pass
