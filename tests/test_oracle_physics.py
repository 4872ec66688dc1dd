"""Analytic validation of the oracle's rigid-body physics (oracle/physics.c).

MuJoCo is not available here, so absolute parity with mj_step is UNPINNED; these
tests pin the restatement to closed-form mechanics instead: the cart-pole equations of motion,
energy conservation, free fall of a free body, the pendulum period, soft joint-limit equilibrium,
and an independent (Jacobian-based, numpy) mass matrix."""
import math
import os
import tempfile

import numpy as np
import pytest

from mujoco_mpc_amd import mjcf
from mujoco_mpc_amd.cstructs import PackedModel
from oracle import pyoracle


def _model(xml):
    d = tempfile.mkdtemp()
    p = os.path.join(d, "m.xml")
    open(p, "w").write(xml)
    return mjcf.load_xml(p)


def test_cartpole_matches_lagrangian(cartpole):
    pm = cartpole.packed_model()
    ph = pyoracle.Physics(pm)
    mc, mp, l, g = 1.0, 0.1, 0.5, 9.81
    Ic = cartpole.model.body_inertia[2][0]
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.uniform([-1.5, -3], [1.5, 3]); v = rng.uniform(-2, 2, 2); u = rng.uniform(-1, 1)
        ph.set_state(q, v); ph.set_ctrl([u]); ph.forward()
        th = q[1]
        M = np.array([[mc + mp, mp * l * math.cos(th)], [mp * l * math.cos(th), Ic + mp * l * l]])
        bias = np.array([-mp * l * math.sin(th) * v[1] ** 2, -mp * g * l * math.sin(th)])
        f = np.array([10 * u - 1e-4 * v[0], -1e-4 * v[1]]) - bias
        assert np.allclose(ph.get("M").reshape(2, 2), M, atol=1e-14)
        assert np.allclose(ph.get("qacc"), np.linalg.solve(M, f), rtol=1e-12, atol=1e-12)
        assert np.allclose(ph.get("M").reshape(2, 2), mjcf.mass_matrix(cartpole.model, q), atol=1e-14)


DOUBLE_PENDULUM = """
<mujoco><option timestep="0.0005"><flag contact="disable"/></option>
<worldbody><body name="a" pos="0 0 2"><joint name="j1" type="hinge" axis="0 1 0"/>
<geom type="capsule" fromto="0 0 0 0.3 0 -0.5" size="0.03" mass="0.7"/>
<body name="b" pos="0.3 0 -0.5" euler="0 20 10"><joint name="j2" type="hinge" axis="1 1 0" pos="0 0.02 0"/>
<geom type="box" size="0.05 0.1 0.2" pos="0 0.05 -0.2" mass="0.4"/>
<body name="c" pos="0 0 -0.4"><joint name="j3" type="slide" axis="0 0 1" stiffness="30" springref="0.1"/>
<geom type="sphere" size="0.06" mass="0.2"/></body></body></body></worldbody></mujoco>"""


def test_energy_conservation_3dof_tree():
    fm = _model(DOUBLE_PENDULUM)
    ph = pyoracle.Physics(PackedModel(fm))
    ph.set_state([0.4, -0.7, 0.05], [0.5, -1.0, 0.2]); ph.forward()
    e0 = ph.get("energy").sum()
    for _ in range(4000):
        ph.step()
    ph.forward()
    e1 = ph.get("energy").sum()
    assert abs(e1 - e0) < 2e-3 * (abs(e0) + 1), (e0, e1)   # semi-implicit Euler, h = 0.5 ms, 2 s
    q = ph.get("qpos")
    assert np.allclose(ph.get("M").reshape(3, 3), mjcf.mass_matrix(fm, q), atol=1e-12)


def test_pendulum_period():
    fm = _model("""<mujoco><option timestep="0.0002"><flag contact="disable"/></option><worldbody>
      <body pos="0 0 1"><joint type="hinge" axis="0 1 0"/><geom type="sphere" size="0.001" pos="0 0 -0.8" mass="1"/></body>
      </worldbody></mujoco>""")
    ph = pyoracle.Physics(PackedModel(fm))
    ph.set_state([0.05], [0.0])
    t_cross, prev = [], 0.05
    for k in range(20000):
        ph.step()
        q = ph.get("qpos")[0]
        if prev > 0 >= q:
            t_cross.append((k + 1) * 0.0002)
        prev = q
    period = t_cross[1] - t_cross[0]
    assert abs(period - 2 * math.pi * math.sqrt(0.8 / 9.81)) < 2e-3


def test_free_body_free_fall_and_spin():
    fm = _model("""<mujoco><option timestep="0.001"><flag contact="disable"/></option><worldbody>
      <body pos="0 0 3"><freejoint/><geom type="box" size="0.1 0.2 0.3" mass="2"/></body></worldbody></mujoco>""")
    assert fm.nq == 7 and fm.nv == 6
    ph = pyoracle.Physics(PackedModel(fm))
    w = np.array([0.0, 0.0, 2.0])  # spin about a principal axis: constant angular velocity
    ph.set_state([0, 0, 3, 1, 0, 0, 0], [1.0, 0, 0, *w])
    n = 500
    for _ in range(n):
        ph.step()
    q, v = ph.get("qpos"), ph.get("qvel")
    h = 0.001
    assert abs(q[0] - 1.0 * n * h) < 1e-12
    assert abs(q[2] - (3 - 9.81 * h * h * n * (n + 1) / 2)) < 1e-10   # semi-implicit Euler closed form
    assert np.allclose(v[3:], w, atol=1e-12)
    ang = 2.0 * n * h
    assert np.allclose(q[3:], [math.cos(ang / 2), 0, 0, math.sin(ang / 2)], atol=1e-10)


def test_joint_limit_soft_equilibrium(particle):
    """A constant push against a limited slide joint settles where the constraint force balances it."""
    pm = particle.packed_model(planning=False)
    ph = pyoracle.Physics(pm)
    ph.set_state([0.285, 0.0], [0.0, 0.0])
    for _ in range(3000):
        ph.set_ctrl([1.0, 0.0])
        ph.step()
    ph.set_ctrl([1.0, 0.0]); ph.forward()
    q, v = ph.get("qpos"), ph.get("qvel")
    assert q[0] > 0.29 and q[0] < 0.30 and abs(v[0]) < 1e-6
    assert ph.get("nefc")[0] == 1
    assert abs(ph.get("qfrc_constraint")[0] + 1.0) < 1e-6          # balances gear*ctrl = 1 N
    assert ph.get("efc_force")[0] > 0


def test_limit_inactive_inside_range(particle):
    ph = pyoracle.Physics(particle.packed_model())
    ph.set_state([0.1, -0.2], [0, 0]); ph.set_ctrl([0.3, -0.4]); ph.forward()
    assert ph.get("nefc")[0] == 0
    assert np.allclose(ph.get("qacc"), np.array([0.3, -0.4]) / 0.3, rtol=1e-13)


def test_euler_implicit_damping(particle):
    """(M + h B) qacc' = f: one step of the damped point mass has a closed form."""
    pm = particle.packed_model(planning=False)
    ph = pyoracle.Physics(pm)
    ph.set_state([0, 0], [0.2, -0.1]); ph.set_ctrl([0.5, 0.0]); ph.step()
    h, m, b = 0.01, 0.3, 1.0
    v0 = np.array([0.2, -0.1]); f = np.array([0.5, 0.0]) - b * v0
    v1 = v0 + h * f / (m + h * b)
    assert np.allclose(ph.get("qvel"), v1, rtol=1e-13)
    assert np.allclose(ph.get("qpos"), h * v1, rtol=1e-13)
    assert abs(ph.get("time")[0] - 0.01) < 1e-15


def test_unsupported_features_are_rejected():
    fm = _model("""<mujoco><option integrator="implicit"><flag contact="disable"/></option><worldbody><body><joint type="hinge"/>
      <geom type="sphere" size="0.1"/></body></worldbody></mujoco>""")
    with pytest.raises(NotImplementedError):
        pyoracle.Physics(PackedModel(fm))


def test_implicitfast_steps_like_euler_on_a_damping_only_model_and_is_refused_otherwise():
    """include/mjpcx.h: MJPCX_INT_IMPLICITFAST is accepted where MuJoCo's implicit-in-velocity update equals mj_Euler's (joint damping is
    the only velocity-dependent smooth force); an actuator with a velocity term in its bias, or MJPCX_INT_IMPLICIT, is rejected"""
    import numpy as np
    from mujoco_mpc_amd.task import load_task
    from oracle import pyoracle
    t = load_task("Cartpole")
    N, H, P = 3, 12, 3
    times = np.arange(P) * 0.01 * (H - 1) / (P - 1)
    nodes = np.random.default_rng(0).normal(0, 0.3, (N, P, 1))
    out = []
    for integ in (0, 3):
        pm = t.packed_model(); pm.struct.integrator = integ
        out.append(pyoracle.rollout_batch(pm, t.packed(), [0.3, 2.7, -0.4, 0.9], 0.0, None, N, H, P, 2, times, nodes, num_threads=1))
    assert np.array_equal(out[0]["states"], out[1]["states"]) and np.array_equal(out[0]["total_return"], out[1]["total_return"])
    # refused: plain `implicit`; implicitfast with an AFFINE bias that has a velocity term; implicitfast with eulerdamp disabled (mj_implicit
    # ignores that flag, o_euler honours it). A velocity coefficient under biastype none is not read by MuJoCo and does not count.
    for integ, kv, affine, flags in ((2, 0.0, 0, 0), (3, -0.5, 1, 0), (3, 0.0, 0, 1 << 14)):
        pm = t.packed_model(); pm.struct.integrator = integ
        pm.struct.disableflags |= flags
        np.ctypeslib.as_array(pm.struct.actuator_biasprm, (3 * t.model.nu,))[2] = kv
        np.ctypeslib.as_array(pm.struct.actuator_biastype, (t.model.nu,))[0] = affine
        try:
            pyoracle.Physics(pm)
        except NotImplementedError:
            continue
        raise AssertionError("accepted")
    pm = t.packed_model(); pm.struct.integrator = 3
    np.ctypeslib.as_array(pm.struct.actuator_biasprm, (3 * t.model.nu,))[2] = -0.5
    np.ctypeslib.as_array(pm.struct.actuator_biastype, (t.model.nu,))[0] = 0
    pyoracle.Physics(pm)   # accepted
