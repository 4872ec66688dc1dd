"""The oracle's mj_RungeKutta(4) (oracle/physics.c o_rk4) against (a) an independent numpy RK4 built from the oracle's own
mj_forward, stage by stage, in MuJoCo's order of operations (engine_forward.c mj_RungeKutta); (b) the method's defining
properties: fourth-order convergence on a slide/hinge tree and near-conservation of energy where Euler drifts."""
import math

import numpy as np
import pytest

from mujoco_mpc_amd.cstructs import PackedModel
from oracle import pyoracle
from test_oracle_physics import DOUBLE_PENDULUM, _model

RK4 = 1

TUMBLER = """
<mujoco><option timestep="0.002" gravity="0 0 -9.81"><flag contact="disable"/></option>
<worldbody><body name="a" pos="0 0 1"><freejoint/>
<geom type="box" size="0.05 0.1 0.2" pos="0.02 0 0.01" euler="10 20 30" mass="0.9"/>
<body name="b" pos="0.1 0 0.2"><joint type="ball"/><geom type="capsule" fromto="0 0 0 0.2 0.1 0" size="0.02" mass="0.3"/>
<body name="c" pos="0.2 0.1 0"><joint type="hinge" axis="0 0 1" stiffness="3"/><geom type="sphere" pos="0.1 0 0" size="0.04" mass="0.2"/></body>
</body></body></worldbody></mujoco>"""


def _quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _integrate_pos(fm, q, v, h):
    """mj_integratePos over the flat model: free = 3 translations + quaternion, ball = quaternion"""
    q = q.copy()
    for j in range(len(fm.jnt_type)):
        qa, da, jt = fm.jnt_qposadr[j], fm.jnt_dofadr[j], fm.jnt_type[j]
        if jt == 0:
            q[qa:qa + 3] += h * v[da:da + 3]
            qa, da = qa + 3, da + 3
        if jt in (0, 1):
            w = v[da:da + 3]
            n = np.linalg.norm(w)
            ax = w / n if n > 1e-15 else np.array([1.0, 0, 0])
            qr = np.concatenate([[math.cos(h * n / 2)], math.sin(h * n / 2) * ax])
            q0 = q[qa:qa + 4] / np.linalg.norm(q[qa:qa + 4])
            q[qa:qa + 4] = _quat_mul(q0, qr)
        else:
            q[qa] += h * v[da]
    return q


def _numpy_rk4_step(fm, ph, q, v, t, h):
    A = [[0.5], [0, 0.5], [0, 0, 1.0]]
    B = [1 / 6, 1 / 3, 1 / 3, 1 / 6]
    ph.set_state(q, v, t); ph.forward()
    F = [(v.copy(), ph.get("qacc")[:len(v)].copy())]
    for i in range(1, 4):
        dv = sum(A[i - 1][j] * F[j][0] for j in range(i)); da = sum(A[i - 1][j] * F[j][1] for j in range(i))
        qi, vi = _integrate_pos(fm, q, dv, h), v + h * da
        ph.set_state(qi, vi, t); ph.forward()
        F.append((vi.copy(), ph.get("qacc")[:len(v)].copy()))
    dv = sum(B[j] * F[j][0] for j in range(4)); da = sum(B[j] * F[j][1] for j in range(4))
    return _integrate_pos(fm, q, dv, h), v + h * da


def test_rk4_step_equals_a_numpy_rk4_over_the_forward_pass():
    fm = _model(TUMBLER)
    pm = PackedModel(fm, integrator=RK4)
    ph, ph2 = pyoracle.Physics(pm), pyoracle.Physics(pm)
    rng = np.random.default_rng(1)
    q = np.array(fm.qpos0, float); q[7:11] = [0.8, 0.2, -0.5, 0.1]; q[7:11] /= np.linalg.norm(q[7:11]); q[11] = 0.3
    v = rng.normal(0, 2.0, 10)
    t = 0.0
    ph.set_state(q, v, t)
    for _ in range(25):
        ph.step()
        q, v = _numpy_rk4_step(fm, ph2, q, v, t, 0.002)
        t += 0.002
        np.testing.assert_allclose(ph.get("qpos")[:12], q, rtol=0, atol=1e-12)
        np.testing.assert_allclose(ph.get("qvel")[:10], v, rtol=0, atol=1e-11)
    assert abs(ph.get("time")[0] - t) < 1e-15


def _integrate(fm, integrator, h, T, q, v):
    ph = pyoracle.Physics(PackedModel(fm, timestep=h, integrator=integrator))
    ph.set_state(q, v)
    for _ in range(int(round(T / h))):
        ph.step()
    return np.concatenate([ph.get("qpos")[:len(q)], ph.get("qvel")[:len(v)]])


def test_rk4_is_fourth_order_on_a_hinge_slide_tree():
    fm = _model(DOUBLE_PENDULUM)
    q, v = [0.4, -0.7, 0.05], [0.5, -1.0, 0.2]
    ref = _integrate(fm, RK4, 0.00025, 0.2, q, v)
    err = [np.abs(_integrate(fm, RK4, h, 0.2, q, v) - ref).max() for h in (0.02, 0.01, 0.005)]
    assert 11 < err[0] / err[1] < 22 and 11 < err[1] / err[2] < 22, err
    e_euler = np.abs(_integrate(fm, 0, 0.005, 0.2, q, v) - ref).max()
    assert e_euler > 1e3 * err[2]


def test_rk4_keeps_the_energy_of_a_tumbling_free_body():
    fm = _model(TUMBLER)
    drift = {}
    for integ in (0, RK4):
        ph = pyoracle.Physics(PackedModel(fm, integrator=integ))
        q = np.array(fm.qpos0, float)
        v = np.array([0.3, -0.2, 1.0, 3.0, -2.0, 1.0, 1.0, 2.0, -1.0, 4.0])
        ph.set_state(q, v); ph.forward()
        e0 = ph.get("energy").sum()
        for _ in range(500):
            ph.step()
        ph.forward()
        drift[integ] = abs(ph.get("energy").sum() - e0)
    assert drift[RK4] < 1e-4 and drift[RK4] < 1e-3 * drift[0], drift


def test_unsupported_integrators_are_still_refused():
    fm = _model(TUMBLER)
    with pytest.raises(Exception):
        pyoracle.Physics(PackedModel(fm, integrator=2))
