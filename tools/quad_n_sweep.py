#!/usr/bin/env python3
"""Launch time of the QuadrupedFlat rollout against the batch size, for the quad kernel and for the wavefront-per-candidate kernel
(MJPCX_NO_QUAD=1): where the hand-over between the two belongs when a rank holds a small share of the batch (strong scaling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task

t = load_task("QuadrupedFlat"); t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
state = np.concatenate([t.model.keyframes["home"]["qpos"], np.zeros(18)])
mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
H, P = 100, 3
times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
ns = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_SAMPLING, std0=0.1)
for label, env in (("quad", {}), ("tree", {"MJPCX_NO_QUAD": "1"})):
    os.environ.pop("MJPCX_NO_QUAD", None)
    os.environ.update(env)
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.0, mocap)
    for N in (256, 512, 1024, 2048, 4096, 8192, 16384):
        best = 1e9
        for rep in range(3):
            ctx.sync(); t0 = time.time()
            ctx.rollout_noise(N, H, 0, times, np.zeros((P, 12)), ns)
            ctx.sync(); best = min(best, time.time() - t0)
        print("%s N = %5d: %7.2f ms  (%.0f k rollouts/s)" % (label, N, 1e3 * best, N / best / 1e3), flush=True)
    ctx.close()
