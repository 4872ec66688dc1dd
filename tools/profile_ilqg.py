#!/usr/bin/env python3
"""Times the iLQG dense kernels at the Quadruped iLQG shape of BASELINE configs[4] (n = 36, m = 12, T = 36)
on random SPD problems (the physics-independent part of that config), and the small-model pipeline on Cartpole."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mujoco_mpc_amd import capi  # noqa: E402
from mujoco_mpc_amd.task import load_task  # noqa: E402
from test_gpu_ilqg import random_lq  # noqa: E402

task = load_task("Cartpole")
ctx = capi.Context(task.packed_model(), task.packed(), 0, 64)
for n, m, T in ((4, 1, 101), (36, 12, 36), (48, 16, 36)):
    prob = random_lq(n, m, T, 0)
    for lim in (1, 0):
        ms = []
        for _ in range(10):
            out = ctx.backward_pass(0.3, 0, lim, *prob)
            ms.append(out["kernel_ms"])
        flops = (T - 1) * 2.0 * (2 * n ** 3 + 3 * n * n * m + 2 * n * m * m + m ** 3 / 3 + n * n * m)
        print(f"backward_pass n={n} m={m} T={T} limits={lim}: kernel {np.median(ms) * 1e3:.1f} us "
              f"({np.median(ms) * 1e3 / (T - 1):.2f} us/step, {flops / (np.median(ms) * 1e-3) / 1e9:.2f} GFLOP/s of Riccati flops)")
# small-model pipeline pieces
H = 101
times = np.arange(H) * 0.01
states = np.zeros((H, 4)); states[:, 1] = 0.3
actions = np.zeros((H, 1))
for name, fn in (("transition_fd", lambda: ctx.transition_fd(times, states, actions)),):
    fn()
    t0 = time.perf_counter()
    for _ in range(20):
        A, B, C, D = fn()
    print(f"{name} T={H}: {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per call (host-inclusive)")
res = np.zeros((H, 4))
ctx.cost_derivatives(res, C, D)
t0 = time.perf_counter()
for _ in range(20):
    ctx.cost_derivatives(res, C, D)
print(f"cost_derivatives T={H}: {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per call (host-inclusive)")

# ---- BASELINE configs[4]: iLQG on the Quadruped (T = 36, 10 line-search rollouts, forward differences)
from mujoco_mpc_amd.planners import GpuILQGPlanner, State  # noqa: E402
qt = load_task("QuadrupedFlat")
qt.transition(0.0)
Hq = qt.planning_steps()
pl = GpuILQGPlanner()
pl.initialize(qt.model, qt); pl.allocate(); pl.reset(Hq)
st = State(qt.model)
home = qt.model.keyframes["home"]["qpos"]
st.set(home, np.zeros(18), mocap_pos=[[0.3, 0, 0.26], [-2.5, 0, 0]], mocap_quat=[[1, 0, 0, 0], [1, 0, 0, 0]], time=0.0)
pl.set_state(st)
for k in range(4):
    t0 = time.perf_counter()
    pl.optimize_policy(Hq)
    el = (time.perf_counter() - t0) * 1e3
    print(f"quadruped iLQG iteration {k}: {el:.1f} ms total; return {pl.policy.trajectory.total_return:.4f}; " +
          ", ".join(f"{n} {v / 1e3:.1f} ms" for n, v in pl.timers.items()))
