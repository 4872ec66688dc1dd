#!/usr/bin/env python3
"""Writes tests/golden/*.npz: seeded inputs and the CPU oracle's outputs for the rollout path and the iLQG pieces.

The reference itself cannot run in this environment (MuJoCo is a fetch-at-configure dependency), so
these fixtures are NOT reference outputs: they freeze the oracle (whose mjpc-owned arithmetic is pinned by the
reference's own known-answer tests, see tests/test_spline.py, test_norm_cost.py, test_oracle_riccati.py,
test_oracle_rollout.py) so that (a) the oracle cannot drift silently and (b) the GPU box can check the HIP path against
committed vectors without executing anything under oracle/. Re-run after an intentional oracle change:
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mujoco_mpc_amd import capi  # noqa: E402
from mujoco_mpc_amd.task import load_task  # noqa: E402
from oracle import pyoracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

ROLLOUT_CASES = {
    # name: (task, state, time, mocap, N, H, P, interp, node-time span, seed)
    "cartpole_cubic": ("Cartpole", [0.3, 2.7, -0.4, 0.9], 0.25, None, 24, 64, 10, capi.SPLINE_CUBIC, 0.63, 11),
    "cartpole_zero_limit": ("Cartpole", [1.75, 3.0, 1.5, 0.0], 0.0, None, 16, 48, 3, capi.SPLINE_ZERO, 0.47, 12),
    "particle_linear": ("Particle", [0.05, -0.1, 0.2, 0.0], 0.1, [0.15, -0.1, 0.01, 1, 0, 0, 0], 20, 40, 5, capi.SPLINE_LINEAR, 0.39, 13),
}


# contact models: (task, transition mode or None, N, H, P, interp, noise std, seed, GPU tolerance); the state is the task's
# start state (A1: home keyframe; humanoid: first frame of the Walk clip, mocap markers from Transition)
CONTACT_CASES = {
    "quadruped_trot_zero": ("QuadrupedFlat", None, 4, 30, 3, capi.SPLINE_ZERO, 0.4, 21, 1e-6),
    "humanoid_walk_cubic": ("HumanoidTrack", 9, 3, 24, 6, capi.SPLINE_CUBIC, 0.3, 22, 1e-6),
}


def prepare_contact_task(tname, mode):
    """task with its Transition applied at time 0, start state and mocap (7 per mocap body) -- shared with tests/test_golden.py"""
    task = load_task(tname)
    if tname == "QuadrupedFlat":
        task.parameters[task.parameter_index("select_Gait")] = 2  # Trot: the gait terms of the residual are active
        task.transition(0.0)
        state = np.concatenate([task.model.keyframes["home"]["qpos"], np.zeros(task.model.nv)])
        mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0])
    else:
        e = task.transition(0.0, mode=mode)
        state = np.concatenate([e["qpos"], e["qvel"]])
        mocap = np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(e["mocap_pos"]).reshape(-1, 3)])
    return task, state, mocap


def contact_case(name):
    tname, mode, N, H, P, interp, std, seed, tol = CONTACT_CASES[name]
    task, state, mocap = prepare_contact_task(tname, mode)
    dt = task.model.get_number("agent_timestep", task.model.timestep)
    times = np.arange(P) * ((H - 1) * dt / (P - 1))
    nodes = np.clip(np.random.default_rng(seed).normal(0, std, (N, P, task.model.nu)), -1, 1)
    ref = pyoracle.rollout_batch(task.packed_model(), task.packed(), state, 0.0, mocap, N, H, P, interp, times, nodes, num_threads=1)
    assert not ref["failure"].any()
    inputs = dict(task=tname, state=state, time=0.0, mocap=mocap, N=N, H=H, P=P, interp=interp, times=times, nodes=nodes,
                  mode=-1 if mode is None else mode, gpu_tol=tol)
    return inputs, ref


def rollout_case(name):
    tname, state, time, mocap, N, H, P, interp, span, seed = ROLLOUT_CASES[name]
    task = load_task(tname)
    rng = np.random.default_rng(seed)
    times = time + np.arange(P) * (span / max(P - 1, 1))
    nodes = np.clip(rng.normal(0, 0.6, (N, P, task.model.nu)), -1, 1)
    ref = pyoracle.rollout_batch(task.packed_model(), task.packed(), state, time, mocap, N, H, P, interp, times, nodes,
                                 num_threads=1)
    inputs = dict(task=tname, state=np.asarray(state, float), time=time,
                  mocap=np.zeros(0) if mocap is None else np.asarray(mocap, float), N=N, H=H, P=P, interp=interp,
                  times=times, nodes=nodes)
    return inputs, ref


def ilqg_case():
    from test_gpu_ilqg import random_lq
    n, m, T = 6, 2, 12
    prob = random_lq(n, m, T, 5)
    out = {}
    for lim in (0, 1):
        for reg in (0, 1, 2):
            r = pyoracle.riccati(n, m, T, 0.3, reg, lim, *prob)
            assert r["ok"]
            for k in ("Vx", "Vxx", "K", "du", "dV"):
                out[f"lim{lim}_reg{reg}_{k}"] = r[k]
    names = ("A", "B", "cx", "cu", "cxx", "cxu", "cuu", "actions", "limits")
    return dict(n=n, m=m, T=T, mu=0.3, **{k: np.asarray(v) for k, v in zip(names, prob)}), out


def main():
    os.makedirs(OUT, exist_ok=True)
    pyoracle.build()
    for name in ROLLOUT_CASES:
        inputs, ref = rollout_case(name)
        np.savez_compressed(os.path.join(OUT, f"rollout_{name}.npz"), **{f"in_{k}": v for k, v in inputs.items()},
                            **{f"out_{k}": np.asarray(v) for k, v in ref.items()})
    for name in CONTACT_CASES:
        inputs, ref = contact_case(name)
        np.savez_compressed(os.path.join(OUT, f"rollout_{name}.npz"), **{f"in_{k}": v for k, v in inputs.items()},
                            **{f"out_{k}": np.asarray(v) for k, v in ref.items()})
    inputs, out = ilqg_case()
    np.savez_compressed(os.path.join(OUT, "riccati_random_lq.npz"), **{f"in_{k}": v for k, v in inputs.items()},
                        **{f"out_{k}": v for k, v in out.items()})
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
