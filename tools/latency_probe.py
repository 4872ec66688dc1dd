#!/usr/bin/env python3
"""Latency of a FEW QuadrupedFlat rollouts (the regime of the iLQG phases, configs[4]: 1 and 10 candidates x 36 steps): the registered
kernel (model image in LDS) against the generic one (model through the caches), fp64 and fp32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task

t = load_task("QuadrupedFlat"); t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
state = np.concatenate([t.model.keyframes["home"]["qpos"], np.zeros(18)])
mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
H, P = 36, 3
times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
ns = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_SAMPLING, std0=0.1)
for prec in (64, 32):
    for label, env in (("registered", {}), ("generic", {"MJPCX_NO_LDS_MODEL": "1"}), ("quad", {"MJPCX_QUAD_MIN_N": "0"})):
        if label == "quad" and prec != 64:
            continue
        os.environ.pop("MJPCX_NO_LDS_MODEL", None)
        os.environ.pop("MJPCX_QUAD_MIN_N", None)
        os.environ.update(env)
        ctx = capi.Context(pm, pt, 0, prec)
        ctx.set_state(state, 0.0, mocap)
        for N in (1, 10, 16, 64, 256):
            best = 1e9
            for rep in range(5):
                ctx.sync(); t0 = time.time()
                ctx.rollout_noise(N, H, 0, times, np.zeros((P, 12)), ns)
                ctx.sync(); best = min(best, time.time() - t0)
            print("fp%d %-10s %-40s N = %3d: %6.2f ms = %.3f ms per step" % (prec, label, ctx.kernel_name[:40], N, 1e3 * best, 1e3 * best / H), flush=True)
        ctx.close()
