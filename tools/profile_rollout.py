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
ap.add_argument("--interp", type=int, default=2)
ap.add_argument("--std", type=float, default=0.5)
a = ap.parse_args()
task = load_task(a.task)
ctx = capi.Context(task.packed_model(), task.packed(), 0, a.precision)
P = int(task.model.get_number("sampling_spline_points", 10))
dt = task.model.get_number("agent_timestep", task.model.timestep)
times = np.arange(P) * ((a.horizon - 1) * dt / (P - 1))
home = task.model.keyframes.get("home")
state = np.concatenate([home["qpos"], home["qvel"]]) if home else np.zeros(task.model.nq + task.model.nv)
mocap = None
if a.task == "HumanoidTrack":  # first frame of the Walk clip; the markers come from the task's Transition
    e = task.transition(0.0, mode=9)
    state = np.concatenate([e["qpos"], e["qvel"]])
    mocap = np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(e["mocap_pos"]).reshape(-1, 3)])
elif hasattr(task, "transition"):
    task.transition(0.0)
if hasattr(task, "transition"):
    ctx.set_task_params(task.weight, task.norm_parameter, task.parameters, task.risk)
    ctx.set_residual_state(task.residual_int, task.residual_real)
ctx.set_state(state, 0.0, mocap)
ctx.timing_reset()
for k in range(a.launches):
    ctx.rollout_noise(a.n, a.horizon, a.interp, times, np.zeros((P, task.model.nu)),
                      capi.make_noise_spec(seed=0, iteration=k, std0=a.std))
ms, n = ctx.timing_read()
ret, fail = ctx.returns()
print(f"returns: min {ret.min():.4f} median {np.median(ret):.4f} failures {int(fail.sum())}")
print(f"{ctx.kernel_name}: {n} launches, avg {ms / n * 1e3:.1f} us, {a.n * a.horizon / (ms / n * 1e-3) / 1e6:.2f} M steps/s, "
      f"{ctx.algorithmic_bytes(a.horizon, P) * a.n / (ms / n * 1e-3) / 1e9:.1f} GB/s algorithmic")
