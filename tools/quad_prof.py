#!/usr/bin/env python3
"""K launches of the north-star batch (QuadrupedFlat, Predictive-Sampling noise, 16384 x 100, fp64) and nothing else: the rocprofv3 target
for the quad kernel (kernel trace, PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
t = load_task("QuadrupedFlat"); t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
state = np.concatenate([t.model.keyframes["home"]["qpos"], np.zeros(18)])
mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
H, P = 100, 3
times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
ctx = capi.Context(pm, pt, 0, 64)
ctx.set_state(state, 0.0, mocap)
ns = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_SAMPLING, std0=0.04)
for k in range(K):
    ctx.rollout_noise(N, H, 0, times, np.zeros((P, 12)), ns)
ctx.sync()
print("done", ctx.kernel_name)
