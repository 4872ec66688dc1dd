"""Port of the reference's rollout tests: mjpc/test/agent/rollout_test.cc:67-153 (PD policy on the
particle, residual[t] pairs with states[t]) and mjpc/test/agent/trajectory_test.cc (buffers)."""
import numpy as np

from mujoco_mpc_amd import capi
from oracle import pyoracle


def test_particle_pd_rollout(particle_copy):
    t = particle_copy
    pm, pt = t.packed_model(planning=False), t.packed()     # the test uses the model's own dt = 0.01
    ph = pyoracle.Physics(pm)
    horizon = 100
    mocap = np.array([0.25, 0, 0.01, 1, 0, 0, 0.0])
    tr = pyoracle.rollout_pd(pm, pt, ph, np.zeros(4), 0.0, mocap, horizon, [0.1, 0.1], [0.0, 0.0], 10.0, 2.5)
    assert np.abs(tr.states[-1, :2] - 0.1).sum() < 0.1       # rollout_test.cc:137
    assert np.abs(tr.states[-1, 2:]).sum() < 0.1             # :138
    assert np.abs(tr.states - tr.residual).sum() < 1e-5      # :141-145: residual == states
    assert not tr.failure
    assert np.allclose(tr.times, np.arange(horizon) * 0.01, atol=1e-12)
    assert np.array_equal(tr.actions[-1], tr.actions[-2])    # trajectory.cc:190-196
    assert np.allclose(tr.trace[:, :2], tr.states[:, :2]) and np.allclose(tr.trace[:, 2], 0.01)


def test_return_is_mean_cost(cartpole):
    t = cartpole
    pm, pt = t.packed_model(), t.packed()
    ph = pyoracle.Physics(pm)
    sp = pyoracle.Spline(1, capi.SPLINE_CUBIC)
    for k, v in enumerate([0.3, -0.8, 1.0, 0.2]):
        sp.add_node(0.1 * k, [v])
    H = 40
    tr = pyoracle.rollout_spline(pm, pt, ph, [0.2, 2.8, 0, 0], 0.5, None, H, sp)
    assert abs(tr.total_return - tr.costs.sum() / H) < 1e-14          # trajectory.cc:312-326
    for k in range(H):
        assert abs(tr.costs[k] - pyoracle.cost_value(pt, tr.residual[k])) < 1e-15
    assert abs(tr.times[0] - 0.5) < 1e-15 and abs(tr.times[1] - 0.51) < 1e-15
    assert np.all(np.abs(tr.actions) <= 1.0)
    # residual definition, cartpole.cc:36-49
    assert np.allclose(tr.residual[:, 0], np.cos(tr.states[:, 1]) - 1)
    assert np.allclose(tr.residual[:, 1], tr.states[:, 0])
    assert np.allclose(tr.residual[:, 2], tr.states[:, 3])
    assert np.allclose(tr.residual[:, 3], tr.actions[:, 0])


def test_horizon_one(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    sp = pyoracle.Spline(1, 0)
    sp.add_node(0.0, [0.7])
    tr = pyoracle.rollout_spline(pm, pt, pyoracle.Physics(pm), [0, 0.3, 0, 0], 0.0, None, 1, sp)
    assert tr.actions[0, 0] == 0.0 and not tr.failure           # trajectory.cc:193-195: zero action
    assert abs(tr.total_return - tr.costs[0]) < 1e-15


def test_divergence_sets_failure(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    sp = pyoracle.Spline(1, 0)
    sp.add_node(0.0, [0.0])
    tr = pyoracle.rollout_spline(pm, pt, pyoracle.Physics(pm), [0, 0, 1e12, 0], 0.0, None, 8, sp)
    assert tr.failure and tr.total_return == 1.0e6              # trajectory.cc:169-173, kMaxReturnValue


def test_batch_equals_single(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    N, H, P = 37, 20, 4
    times = np.linspace(0, 0.19, P)
    nodes = np.random.default_rng(1).uniform(-1, 1, (N, P, 1))
    st = [0.1, 3.0, 0.2, -0.4]
    b1 = pyoracle.rollout_batch(pm, pt, st, 0.0, None, N, H, P, 2, times, nodes, num_threads=1)
    b4 = pyoracle.rollout_batch(pm, pt, st, 0.0, None, N, H, P, 2, times, nodes, num_threads=4)
    for k in b1:
        assert np.array_equal(b1[k], b4[k])
    sp = pyoracle.Spline(1, 2)
    for p in range(P):
        sp.add_node(times[p], nodes[5, p])
    tr = pyoracle.rollout_spline(pm, pt, pyoracle.Physics(pm), st, 0.0, None, H, sp)
    assert np.array_equal(tr.states, b1["states"][5]) and tr.total_return == b1["total_return"][5]
