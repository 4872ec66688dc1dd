#!/usr/bin/env python3
"""fp32 A1 rollouts against the fp64 oracle: relative return errors per candidate (tuning aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle
t = load_task("QuadrupedFlat"); t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
state = np.concatenate([t.model.keyframes["home"]["qpos"], np.zeros(18)])
for H in (5, 40, 100):
    N, P = 16, 4
    rng = np.random.default_rng(H)
    times = np.arange(P) * max((H - 1) * 0.01 / (P - 1), 1e-3)
    nodes = np.clip(rng.normal(0, 0.1, (N, P, 12)), -1, 1)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 1, times, nodes, num_threads=8)
    for env in ({}, {"MJPCX_NO_LDS_MODEL": "1"}, {"MJPCX_NO_TREE": "1"}):
        for k in ("MJPCX_NO_LDS_MODEL", "MJPCX_NO_TREE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx = capi.Context(pm, pt, 0, 32)
        ctx.set_state(state, 0.0, mocap)
        ctx.rollout_splines(H, 1, times, nodes)
        ret, fail = ctx.returns()
        rel = np.abs(ret - ref["total_return"]) / np.abs(ref["total_return"])
        print(H, env, ctx.kernel_name[:24], "max rel %.2e median %.2e" % (rel.max(), np.median(rel)), "fails", int(fail.sum()))
        ctx.close()
