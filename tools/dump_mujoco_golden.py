#!/usr/bin/env python3
"""Pins the oracle's physics against the real MuJoCo -- on any machine that has the `mujoco` wheel (this build
environment does not: SURVEY.md F2, DESIGN.md 5 "parity unpinned").

For every model under mujoco_mpc_amd/models/ that MuJoCo can load, rolls a fixed, seeded control sequence through
mj_step from the model's reference pose and writes states / qacc / sensor-independent kinematics to
tests/golden/mujoco_<task>.npz. tests/test_golden.py::test_oracle_against_mujoco_goldens compares the C oracle with any
such file it finds (and is skipped while there is none).

    pip install mujoco==3.1.6      # the reference pins MuJoCo @ 088079ef (CMakeLists.txt:58-61); the closest wheel
    python tools/dump_mujoco_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

try:
    import mujoco
except ImportError:
    raise SystemExit("the `mujoco` wheel is not installed: nothing to dump (see the docstring)")

from mujoco_mpc_amd.task import _REGISTRY, MODELS_DIR  # noqa: E402

STEPS = 200
for name, (rel, _) in _REGISTRY.items():
    path = os.path.join(MODELS_DIR, rel)
    try:
        m = mujoco.MjModel.from_xml_path(path)
    except Exception as e:  # e.g. a user sensor layout MuJoCo rejects
        print(f"{name}: skipped ({e})")
        continue
    # the agent plans at agent_timestep / Euler (mjpc/agent.cc:288-291); the oracle integrates the same way
    for i in range(m.nnumeric):
        if mujoco.mj_id2name(m, mujoco.mjtObj.mjOBJ_NUMERIC, i) == "agent_timestep":
            m.opt.timestep = m.numeric_data[m.numeric_adr[i]]
    d = mujoco.MjData(m)
    key = mujoco.mj_name2id(m, mujoco.mjtObj.mjOBJ_KEY, "home")
    if key >= 0:
        mujoco.mj_resetDataKeyframe(m, d, key)
    rng = np.random.default_rng(0)
    lo, hi = m.actuator_ctrlrange[:, 0], m.actuator_ctrlrange[:, 1]
    ctrl = np.zeros((STEPS, m.nu))
    hold = None
    for t in range(STEPS):
        if t % 10 == 0:
            hold = rng.uniform(lo, hi) * 0.3 if m.nu else np.zeros(0)
        ctrl[t] = hold
    qpos, qvel, qacc, ncon, nefc, xpos = [], [], [], [], [], []
    for t in range(STEPS):
        d.ctrl[:] = ctrl[t]
        qpos.append(d.qpos.copy()); qvel.append(d.qvel.copy())
        mujoco.mj_step(m, d)
        qacc.append(d.qacc.copy()); ncon.append(d.ncon); nefc.append(d.nefc); xpos.append(d.xpos.copy())
    out = os.path.join(ROOT, "tests", "golden", f"mujoco_{name}.npz")
    np.savez_compressed(out, mujoco_version=mujoco.__version__, timestep=m.opt.timestep, ctrl=ctrl, qpos=np.array(qpos),
                        qvel=np.array(qvel), qacc=np.array(qacc), ncon=np.array(ncon), nefc=np.array(nefc), xpos=np.array(xpos),
                        final_qpos=d.qpos.copy(), final_qvel=d.qvel.copy(), body_mass=m.body_mass.copy(),
                        dof_invweight0=m.dof_invweight0.copy(), meaninertia=m.stat.meaninertia)
    print(f"{name}: {out}")
