#!/usr/bin/env python3
"""Bring-up check of rollout_quad_kernel on the GPU: parity of a small batch against the oracle (all six Trajectory buffers), the
share of candidates handed to the fallback kernel (MJPCX_QUAD_STATS=1), and the launch time of the north-star batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MJPCX_QUAD_STATS", "1")
os.environ.setdefault("MJPCX_QUAD_MIN_N", "0")  # (the 64-candidate parity batches would go to the wavefront-per-candidate kernel otherwise)
import numpy as np
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

t = load_task("QuadrupedFlat"); t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
home = t.model.keyframes["home"]["qpos"]
state = np.concatenate([home, np.zeros(18)])
mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
H, P = 100, 3
times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
ctx = capi.Context(pm, pt, 0, 64)
print("kernel:", ctx.kernel_name)
ctx.set_state(state, 0.0, mocap)
for std in (0.04, 0.15):
    N = 64
    ns = capi.make_noise_spec(seed=7, iteration=1, mode=capi.NOISE_SAMPLING, std0=std)
    nominal = np.zeros((P, 12))
    ctx.rollout_noise(N, H, 0, times, nominal, ns)
    ret, fail = ctx.returns()
    nodes = pyoracle.noise_candidates(pm, ns, P, nominal, np.arange(N))
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 0, times, nodes, num_threads=16)
    worst = 0
    for c in range(0, N, 7):
        tr = ctx.fetch_trajectory(c)
        for k in ("states", "actions", "times", "residual", "costs", "trace"):
            a, b = getattr(tr, k), ref[k][c]
            worst = max(worst, float(np.max(np.abs(a - b) / (1 + np.abs(b)))))
    print("std %.2f: returns max rel diff %.3e, trajectory buffers worst %.3e, failures gpu %d oracle %d" % (
        std, np.max(np.abs(ret - ref["total_return"]) / (1 + np.abs(ref["total_return"]))), worst, int((fail != 0).sum()), int(ref["failure"].sum())))
N = 16384
ns = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_SAMPLING, std0=0.04)
nominal = np.zeros((P, 12))
for rep in range(3):
    ctx.sync(); t0 = time.time()
    ctx.rollout_noise(N, H, 0, times, nominal, ns)
    ctx.sync(); t1 = time.time()
    print("N = %d: %.1f ms per launch = %.0f k rollouts/s" % (N, 1e3 * (t1 - t0), N / (t1 - t0) / 1e3))
ret, fail = ctx.returns()
print("failures", int((fail != 0).sum()), "mean return", float(ret.mean()))
if os.environ.get("MJPCX_NO_QUAD") is None:
    os.environ["MJPCX_NO_QUAD"] = "1"
    ctx2 = capi.Context(pm, pt, 0, 64); ctx2.set_state(state, 0.0, mocap)
    print("reference kernel:", ctx2.kernel_name)
    ctx2.rollout_noise(N, H, 0, times, nominal, ns)
    ctx2.sync(); t0 = time.time(); ctx2.rollout_noise(N, H, 0, times, nominal, ns); ctx2.sync(); t1 = time.time()
    r2, f2 = ctx2.returns()
    print("tree kernel: %.1f ms; returns quad vs tree max rel diff %.3e (median %.3e)" % (
        1e3 * (t1 - t0), np.max(np.abs(ret - r2) / (1 + np.abs(r2))), np.median(np.abs(ret - r2) / (1 + np.abs(r2)))))
