#!/usr/bin/env python3
"""Minimal driver for rocprofv3: K launches of the rollout kernel at BASELINE configs[1]
(Cartpole, 4096 candidates, horizon 128, fp64), nothing else on the GPU."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_mpc_amd import capi  # noqa: E402
from mujoco_mpc_amd.task import load_task  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--task", default="Cartpole")
ap.add_argument("-n", type=int, default=4096)
ap.add_argument("--horizon", type=int, default=128)
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--precision", type=int, default=64)
a = ap.parse_args()
task = load_task(a.task)
ctx = capi.Context(task.packed_model(), task.packed(), 0, a.precision)
P = int(task.model.get_number("sampling_spline_points", 10))
dt = task.model.get_number("agent_timestep", task.model.timestep)
times = np.arange(P) * ((a.horizon - 1) * dt / (P - 1))
home = task.model.keyframes.get("home")
state = np.concatenate([home["qpos"], home["qvel"]]) if home else np.zeros(task.model.nq + task.model.nv)
ctx.set_state(state, 0.0)
ctx.timing_reset()
for k in range(a.launches):
    ctx.rollout_noise(a.n, a.horizon, capi.SPLINE_CUBIC, times, np.zeros((P, task.model.nu)),
                      capi.make_noise_spec(seed=0, iteration=k, std0=0.5))
ms, n = ctx.timing_read()
print(f"{ctx.kernel_name}: {n} launches, avg {ms / n * 1e3:.1f} us, "
      f"{ctx.algorithmic_bytes(a.horizon, P) * a.n / (ms / n * 1e-3) / 1e9:.1f} GB/s algorithmic")
