"""Committed fixtures (tests/golden/*.npz, written by tools/make_golden.py from seeded inputs).

CPU leg: the oracle must still reproduce its frozen outputs exactly-ish (1e-12: same code, same compiler flags).
GPU leg (-m gpu): the HIP path is checked against the SAME committed vectors without running anything under oracle/.
Tolerance for the GPU leg as in test_gpu_parity.py: |gpu - golden| <= 1e-9 (1 + |golden|), fp64."""
import glob
import os
import sys

import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROLLOUTS = sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz")))
FIELDS = ("states", "actions", "times", "residual", "costs", "trace")


def load(path):
    z = np.load(path)
    i = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    o = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return i, o


def close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= tol * (1 + np.abs(b)))


def golden_task(i):
    """the task as tools/make_golden.py prepared it (contact-model cases apply the task's Transition first)"""
    if "mode" in i:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        from make_golden import prepare_contact_task
        return prepare_contact_task(str(i["task"]), None if int(i["mode"]) < 0 else int(i["mode"]))[0]
    return load_task(str(i["task"]))


def test_fixtures_present():
    assert len(ROLLOUTS) >= 3 and os.path.exists(os.path.join(GOLDEN, "riccati_random_lq.npz"))


@pytest.mark.parametrize("path", ROLLOUTS, ids=[os.path.basename(p) for p in ROLLOUTS])
def test_oracle_reproduces_golden_rollouts(path):
    from oracle import pyoracle
    i, o = load(path)
    task = golden_task(i)
    mocap = i["mocap"] if i["mocap"].size else None
    ref = pyoracle.rollout_batch(task.packed_model(), task.packed(), i["state"], float(i["time"]), mocap, int(i["N"]), int(i["H"]),
                                 int(i["P"]), int(i["interp"]), i["times"], i["nodes"], num_threads=2)
    assert np.array_equal(ref["failure"], o["failure"])
    for k in FIELDS + ("total_return",):
        assert close(ref[k], o[k], 1e-12), k


def test_oracle_reproduces_golden_riccati():
    from oracle import pyoracle
    i, o = load(os.path.join(GOLDEN, "riccati_random_lq.npz"))
    args = [i[k] for k in ("A", "B", "cx", "cu", "cxx", "cxu", "cuu", "actions", "limits")]
    for lim in (0, 1):
        for reg in (0, 1, 2):
            r = pyoracle.riccati(int(i["n"]), int(i["m"]), int(i["T"]), float(i["mu"]), reg, lim, *args)
            assert r["ok"]
            for k in ("Vx", "Vxx", "K", "du", "dV"):
                assert close(r[k], o[f"lim{lim}_reg{reg}_{k}"], 1e-12), (lim, reg, k)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ROLLOUTS, ids=[os.path.basename(p) for p in ROLLOUTS])
def test_gpu_matches_golden_rollouts(path):
    i, o = load(path)
    task = golden_task(i)
    tol = float(i["gpu_tol"]) if "gpu_tol" in i else 1e-9
    ctx = capi.Context(task.packed_model(), task.packed(), 0, 64)
    ctx.set_state(i["state"], float(i["time"]), i["mocap"] if i["mocap"].size else None)
    ctx.rollout_splines(int(i["H"]), int(i["interp"]), i["times"], i["nodes"])
    ret, fail = ctx.returns()
    assert np.array_equal(fail, o["failure"])
    assert close(ret, o["total_return"], tol)
    for c in range(int(i["N"])):
        tr = ctx.fetch_trajectory(c)
        for k in FIELDS:
            assert close(getattr(tr, k), o[k][c], tol), (k, c)
    ctx.close()


@pytest.mark.gpu
def test_gpu_matches_golden_riccati(cartpole):
    i, o = load(os.path.join(GOLDEN, "riccati_random_lq.npz"))
    args = [i[k] for k in ("A", "B", "cx", "cu", "cxx", "cxu", "cuu", "actions", "limits")]
    ctx = capi.Context(cartpole.packed_model(), cartpole.packed(), 0, 64)
    T = int(i["T"])
    for lim in (0, 1):
        for reg in (0, 1, 2):
            r = ctx.backward_pass(float(i["mu"]), reg, lim, *args)
            assert r["ok"]
            for k in ("Vx", "Vxx", "dV"):
                assert close(r[k], o[f"lim{lim}_reg{reg}_{k}"], 1e-9), (lim, reg, k)
            for k in ("K", "du"):  # the oracle leaves index T-1 as a copy of T-2 too (backward_pass.cc:297-306)
                assert close(r[k][:T - 1], o[f"lim{lim}_reg{reg}_{k}"][:T - 1], 1e-9), (lim, reg, k)
    ctx.close()


def _mujoco_goldens():
    import glob
    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mujoco_*.npz")))


@pytest.mark.skipif(not _mujoco_goldens(), reason="no MuJoCo goldens: tools/dump_mujoco_golden.py needs the mujoco wheel (parity unpinned)")
@pytest.mark.parametrize("path", _mujoco_goldens() or [None])
def test_oracle_against_mujoco_goldens(path):
    """pins the oracle's physics to MuJoCo itself once a golden file exists: compiled constants first (they catch model
    re-authoring errors), then the first steps of the trajectory (contact-rich rollouts diverge later at any precision)"""
    from mujoco_mpc_amd.task import load_task
    from oracle import pyoracle
    g = np.load(path)
    name = os.path.basename(path)[len("mujoco_"):-len(".npz")]
    t = load_task(name)
    m = t.model
    assert np.allclose(m.arrays["body_mass"], g["body_mass"], rtol=1e-9, atol=1e-12)
    assert np.allclose(m.arrays["dof_invweight0"], g["dof_invweight0"], rtol=1e-6)
    assert abs(m.scalars["meaninertia"] - float(g["meaninertia"])) < 1e-6 * abs(float(g["meaninertia"]))
    ph = pyoracle.Physics(t.packed_model())
    ph.set_state(g["qpos"][0], g["qvel"][0], 0.0)
    for k in range(20):
        ph.set_ctrl(g["ctrl"][k])
        ph.step()
        assert np.allclose(ph.get("qacc"), g["qacc"][k], rtol=1e-5, atol=1e-6), k
        assert np.allclose(ph.get("qpos"), g["qpos"][k + 1], rtol=0, atol=1e-7), k


def test_the_mujoco_golden_dump_stays_runnable():
    """tools/dump_mujoco_golden.py is what turns 'parity unpinned' into a pin on the day a MuJoCo wheel is at hand: it must keep compiling,
    say what is missing when the wheel is not there (this image), and name model files that exist."""
    import py_compile
    import subprocess
    import sys
    from mujoco_mpc_amd.task import _REGISTRY, MODELS_DIR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "dump_mujoco_golden.py")
    py_compile.compile(tool, doraise=True)
    for name, (rel, _) in _REGISTRY.items():
        assert os.path.exists(os.path.join(MODELS_DIR, rel)), name
    try:
        import mujoco  # noqa: F401
    except ImportError:
        out = subprocess.run([sys.executable, tool], capture_output=True, text=True, timeout=120)
        assert out.returncode != 0 and "mujoco" in (out.stderr + out.stdout)
