#!/usr/bin/env python3
"""Debug aid: A1 rollouts on the GPU against the CPU oracle, error per step (first divergence shows which stage is off).
usage: python tools/tree_debug.py [N] [H] [scenario: stand|fall]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_mpc_amd import capi  # noqa: E402
from mujoco_mpc_amd.task import load_task  # noqa: E402
from oracle import pyoracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = int(sys.argv[2]) if len(sys.argv) > 2 else 12
scen = sys.argv[3] if len(sys.argv) > 3 else "stand"
MOCAP = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0])
t = load_task("QuadrupedFlat")
t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
q = t.model.keyframes["home"]["qpos"].copy()
v = np.zeros(18)
if scen == "fall":
    q[2] = 0.5
    q[3:7] = [0.9, 0.3, 0.2, 0.1]
    q[3:7] /= np.linalg.norm(q[3:7])
    v[3:6] = [2.0, -1.0, 0.5]
state = np.concatenate([q, v])
P = 3
rng = np.random.default_rng(1)
times = np.arange(P) * max((H - 1) * 0.01 / (P - 1), 1e-3)
nodes = np.clip(rng.normal(0, 0.4, (N, P, 12)), -1, 1)
ctx = capi.Context(pm, pt, 0, 64)
print(ctx.kernel_name)
ctx.set_state(state, 0.0, MOCAP)
ctx.rollout_splines(H, 0, times, nodes)
ret, fail = ctx.returns()
ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, N, H, P, 0, times, nodes, num_threads=8)
print("fail gpu", fail, "raw", [hex(int(x)) for x in ctx.failure_raw], "oracle", ref["failure"])
print("ret gpu", ret, "\nret ora", ref["total_return"])
for c in range(min(N, 3)):
    tr = ctx.fetch_trajectory(c)
    for s in range(H):
        es = np.max(np.abs(tr.states[s] - ref["states"][c][s]))
        er = np.max(np.abs(tr.residual[s] - ref["residual"][c][s]))
        ec = abs(tr.costs[s] - ref["costs"][c][s])
        print(f"cand {c} step {s:3d}: state err {es:.3e} residual err {er:.3e} cost err {ec:.3e}")
        if es > 1e-3:
            break
