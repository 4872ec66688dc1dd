#!/usr/bin/env python3
"""How much co-resident wavefronts of rollout_quad_kernel slow each other: the same kind of batch (Predictive-Sampling noise around the home
pose, H = 100) at 16 candidates per wavefront (MJPCX_QUAD_CPW=16) on 1, 4, 16, 64, 256, 1024 wavefronts. Every wavefront carries 16 candidates
of the same distribution, so the time of a launch is the slowest wavefront's and differences between the rows are interference (or the
tail of the distribution: more wavefronts, a slower slowest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MJPCX_QUAD_MIN_N"] = "0"
os.environ["MJPCX_QUAD_CPW"] = os.environ.get("MJPCX_QUAD_CPW", "16")
import numpy as np
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task

t = load_task("QuadrupedFlat"); t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
state = np.concatenate([t.model.keyframes["home"]["qpos"], np.zeros(18)])
mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
H, P = 100, 3
times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
ns = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_SAMPLING, std0=0.04)
ctx = capi.Context(pm, pt, 0, 64)
ctx.set_state(state, 0.0, mocap)
cpw = int(os.environ["MJPCX_QUAD_CPW"])
for waves in (1, 4, 16, 64, 256, 512, 1024):
    N = waves * cpw
    best = 1e9
    for rep in range(4):
        ctx.sync(); t0 = time.time()
        ctx.rollout_noise(N, H, 0, times, np.zeros((P, 12)), ns)
        ctx.sync(); best = min(best, time.time() - t0)
    print("%4d wavefronts x %2d candidates: %7.2f ms = %.3f ms per step" % (waves, cpw, 1e3 * best, 1e3 * best / H), flush=True)
