"""-m gpu: parity of the HIP rollout path (through the C ABI) against the CPU oracle.

Tolerances (fp64): |gpu - oracle| <= 1e-9 * (1 + |oracle|) on states, actions, times, residual, costs,
trace and total_return for contact-free rollouts up to H = 128. The two paths share no code: the
oracle is plain C (gcc, -ffp-contract=off, Cholesky, libm sin/cos/pow), the kernel is hipcc with FMA
contraction, LDL' and device libm, so bit-equality is not expected; agreement is ~1e-13 in practice.
fp32 kernels: 2e-3 on returns over short horizons.
"""
import numpy as np
import pytest

from conftest import random_nodes
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

pytestmark = pytest.mark.gpu
TOL = 1e-9


def close(a, b, tol=TOL):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= tol * (1 + np.abs(b)))


def compare_batch(task, state, time, mocap, N, H, P, interp, times, nodes, sample=None, precision=64, tol=TOL, pm=None):
    pm, pt = pm or task.packed_model(), task.packed()
    ctx = capi.Context(pm, pt, 0, precision)
    ctx.set_state(state, time, mocap)
    ctx.rollout_splines(H, interp, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, time, mocap, N, H, P, interp, times, nodes, num_threads=8)
    assert np.array_equal(fail, ref["failure"])
    assert close(ret, ref["total_return"], tol), np.abs(ret - ref["total_return"]).max()
    for c in (sample if sample is not None else range(N)):
        tr = ctx.fetch_trajectory(c)
        assert tr.failure == bool(ref["failure"][c])
        if tr.failure:
            continue
        for name in ("states", "actions", "times", "residual", "costs", "trace"):
            assert close(getattr(tr, name), ref[name][c], tol), (name, c)
        assert abs(tr.total_return - ref["total_return"][c]) <= tol * (1 + abs(ref["total_return"][c]))
    ctx.close()
    return ret, ref


@pytest.mark.parametrize("interp", [0, 1, 2])
@pytest.mark.parametrize("P", [1, 2, 3, 10])
def test_cartpole_interpolations(cartpole, interp, P):
    N, H = 70, 48
    times = 0.25 + np.arange(P) * (0.47 / max(P - 1, 1))
    compare_batch(cartpole, [0.3, 2.7, -0.4, 0.9], 0.25, None, N, H, P, interp, times,
                  random_nodes(interp * 10 + P, N, P, 1))


@pytest.mark.parametrize("N", [1, 63, 64, 65, 257])
def test_ragged_batch_sizes(cartpole, N):
    H, P = 16, 4
    compare_batch(cartpole, [0.0, 3.0, 0.0, 0.0], 0.0, None, N, H, P, 2, np.linspace(0, 0.15, P),
                  random_nodes(N, N, P, 1))


@pytest.mark.parametrize("H", [1, 2, 3, 128])
def test_horizons(cartpole, H):
    N, P = 33, 5
    compare_batch(cartpole, [-0.5, 0.4, 1.0, -2.0], 1.5, None, N, H, P, 2, 1.5 + np.linspace(0, 0.01 * max(H - 1, 1), P),
                  random_nodes(H, N, P, 1))


def test_cartpole_slider_limit_active(cartpole):
    """start next to the +1.8 m slider limit and push into it: the soft limit row is active for most lanes"""
    N, H, P = 128, 64, 4
    nodes = np.clip(np.abs(random_nodes(5, N, P, 1)) + 0.3, -1, 1)
    nodes[::7] *= -1  # a few lanes move away: mixed active / inactive rows within a wavefront
    ret, ref = compare_batch(cartpole, [1.79, 0.1, 1.5, 0.0], 0.0, None, N, H, P, 1, np.linspace(0, 0.63, P), nodes)
    assert ref["states"][:, :, 0].max() > 1.8  # the limit really was violated (soft constraint)


def test_particle_limits_and_mocap(particle):
    N, H, P = 96, 40, 6
    mocap = np.array([0.2, -0.1, 0.01, 1, 0, 0, 0.0])
    nodes = random_nodes(11, N, P, 2, scale=1.5)
    ret, ref = compare_batch(particle, [0.27, -0.28, 0.5, -0.5], 0.0, mocap, N, H, P, 2, np.linspace(0, 3.9, P), nodes)
    assert np.abs(ref["states"][:, :, :2]).max() > 0.29  # both slide limits get hit


def test_particle_copy_residual_equals_state(particle_copy):
    N, H, P = 64, 30, 3
    mocap = np.array([0.25, 0, 0.01, 1, 0, 0, 0.0])
    pm, pt = particle_copy.packed_model(), particle_copy.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(np.zeros(4), 0.0, mocap)
    ctx.rollout_splines(H, 1, np.linspace(0, 2.9, P), random_nodes(2, N, P, 2))
    tr = ctx.fetch_trajectory(17)
    assert np.abs(tr.states - tr.residual).sum() < 1e-5       # mjpc/test/agent/rollout_test.cc:141-145
    compare_batch(particle_copy, np.zeros(4), 0.0, mocap, N, H, P, 1, np.linspace(0, 2.9, P), random_nodes(2, N, P, 2))


def test_rollouts_that_fail_midway_match_up_to_the_failing_step(cartpole):
    """a spline whose later node is NaN (zero-order hold): mjWARN_BADCTRL at the step where the node takes over (trajectory.cc:169-173) --
    at different steps for different candidates. The rollout stops there with the return 1e6, and what it recorded before -- states,
    actions, residual, trace up to and including the failing step, costs before it -- is the oracle's."""
    pm, pt = cartpole.packed_model(), cartpole.packed()
    N, H, P = 4, 40, 3
    dt = float(cartpole.model.get_number("agent_timestep", cartpole.model.scalars["timestep"]))  # the planning step of a rollout
    times = [0.0, 12 * dt, 25 * dt]
    nodes = random_nodes(5, N, P, 1)
    nodes[1, 1, 0] = np.nan        # candidate 1 fails when node 1 takes over, candidate 2 at node 2, the others never
    nodes[2, 2, 0] = np.nan
    state = [0.0, 0.3, 0.1, -0.2]
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.0)
    ctx.rollout_splines(H, 0, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, None, N, H, P, 0, times, nodes)
    assert list(ref["failure"]) == [0, 1, 1, 0] and np.array_equal(fail, ref["failure"])
    assert ret[1] == 1.0e6 and ret[2] == 1.0e6 and close(ret, ref["total_return"])
    steps = []
    for c in (1, 2):
        last = int(np.max(np.nonzero(np.abs(ref["states"][c]).sum(axis=1))[0]))   # the failing step: the last row the oracle wrote
        steps.append(last)
        tr = ctx.fetch_trajectory(c)
        for name in ("states", "residual", "trace"):
            assert close(getattr(tr, name)[:last + 1], ref[name][c][:last + 1]), (name, c)
        assert close(tr.actions[:last], ref["actions"][c][:last]) and close(tr.costs[:last], ref["costs"][c][:last])
    assert 8 < steps[0] < steps[1] < H - 5
    ctx.close()


def test_divergent_candidates_flagged(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 64, 12, 2
    nodes = random_nodes(0, N, P, 1)
    ctx.set_state([0.0, 0.0, 1e12, 0.0], 0.0)                 # |qvel| > mjMAXVAL -> mjWARN_BADQVEL at step 0
    ctx.rollout_splines(H, 0, [0.0, 0.1], nodes)
    ret, fail = ctx.returns()
    assert fail.all() and np.all(ret == 1.0e6)                # trajectory.cc:169-173
    ref = pyoracle.rollout_batch(pm, pt, [0.0, 0.0, 1e12, 0.0], 0.0, None, N, H, P, 0, [0.0, 0.1], nodes)
    assert ref["failure"].all() and np.all(ref["total_return"] == 1.0e6)
    nodes[3, 0, 0] = np.nan                                   # mjWARN_BADCTRL for one candidate only
    ctx.set_state([0.0, 0.1, 0.0, 0.0], 0.0)
    ctx.rollout_splines(H, 0, [0.0, 0.1], nodes)
    ret, fail = ctx.returns()
    assert fail[3] == 1 and ret[3] == 1.0e6 and fail.sum() == 1


def test_task_parameters_are_applied(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 64, 24, 3
    nodes, times = random_nodes(4, N, P, 1), np.linspace(0, 0.23, P)
    state = [0.2, 0.5, 0.0, 0.0]
    import copy
    t2 = copy.copy(cartpole)
    t2.weight = [3.0, 0.5, 0.25, 2.0]; t2.norm_parameter = [0.05, 0.2]; t2.parameters = [0.7]; t2.risk = 0.3
    ctx.set_task_params(t2.weight, t2.norm_parameter, t2.parameters, t2.risk)
    ctx.set_state(state, 0.0)
    ctx.rollout_splines(H, 2, times, nodes)
    ret, _ = ctx.returns()
    ref = pyoracle.rollout_batch(pm, t2.packed(), state, 0.0, None, N, H, P, 2, times, nodes)
    assert close(ret, ref["total_return"])
    ref0 = pyoracle.rollout_batch(pm, pt, state, 0.0, None, N, H, P, 2, times, nodes)
    assert not close(ret, ref0["total_return"], 1e-3)


def test_all_norm_types_on_device(particle):
    """every mjpc::Norm value path (norm.cc:50-210) through the kernel's cost evaluation"""
    import copy
    pm = particle.packed_model()
    N, H, P = 64, 10, 2
    nodes, times = random_nodes(9, N, P, 2), np.array([0.0, 0.9])
    mocap = np.array([0.1, 0.05, 0.01, 1, 0, 0, 0.0])
    for ntype, params in [(0, []), (1, [0.1, 2.0]), (2, [0.1]), (3, [0.5]), (5, [2.0]), (6, [0.1]), (7, [0.1, 2.0]), (8, [0.3])]:
        t2 = copy.copy(particle)
        t2.norm = [ntype, 6]
        t2.num_norm_parameter = [len(params), 1]
        t2.norm_parameter = list(params) + [0.2]
        ctx = capi.Context(pm, t2.packed(), 0, 64)
        ctx.set_state([0.05, -0.1, 0.3, 0.2], 0.0, mocap)
        ctx.rollout_splines(H, 1, times, nodes)
        ret, _ = ctx.returns()
        ref = pyoracle.rollout_batch(pm, t2.packed(), [0.05, -0.1, 0.3, 0.2], 0.0, mocap, N, H, P, 1, times, nodes)
        assert close(ret, ref["total_return"], 1e-9), ntype
        ctx.close()


# ------------------------------------------------------------------ device-side noise
def test_device_noise_matches_spec(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 300, 20, 10
    times = np.linspace(0, 0.19, P)
    nominal = np.linspace(-0.3, 0.6, P).reshape(P, 1)
    for std1 in (0.0, 0.9):
        ns = capi.make_noise_spec(seed=42, iteration=5, std0=0.5, std1=std1, candidate_offset=0, nominal_candidate=0)
        ctx.set_state([0.1, 0.2, 0, 0], 0.0)
        ctx.rollout_noise(N, H, 2, times, nominal, ns)
        ref_nodes = pyoracle.noise_candidates(pm, ns, P, nominal, range(N))
        got = np.stack([ctx.fetch_spline(i) for i in (0, 1, 2, 63, 64, 150, 299)])
        assert np.allclose(got, ref_nodes[[0, 1, 2, 63, 64, 150, 299]], rtol=0, atol=1e-13)
        ret, _ = ctx.returns()
        ref = pyoracle.rollout_batch(pm, pt, [0.1, 0.2, 0, 0], 0.0, None, N, H, P, 2, times, ref_nodes)
        assert close(ret, ref["total_return"])


def test_sharding_invariance(cartpole):
    """candidate i of the global batch is the same whichever rank (candidate_offset) rolls it out"""
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 256, 16, 5
    times, nominal = np.linspace(0, 0.15, P), np.zeros((P, 1))
    ctx.set_state([0, 0.3, 0, 0], 0.0)
    ctx.rollout_noise(N, H, 2, times, nominal, capi.make_noise_spec(seed=7, iteration=1, std0=0.4))
    whole, _ = ctx.returns()
    parts = []
    for r in range(4):
        ctx.rollout_noise(N // 4, H, 2, times, nominal,
                          capi.make_noise_spec(seed=7, iteration=1, std0=0.4, candidate_offset=r * N // 4))
        parts.append(ctx.returns()[0])
    assert np.array_equal(np.concatenate(parts), whole)      # bit-identical


def test_cross_entropy_noise_on_device(particle):
    pm, pt = particle.packed_model(), particle.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 128, 10, 3
    times, nominal = np.array([0.0, 0.4, 0.9]), np.full((P, 2), 0.1)
    var = np.linspace(1e-4, 0.09, P * 2)
    ns = capi.make_noise_spec(seed=3, iteration=2, mode=capi.NOISE_CROSS_ENTROPY, std0=0.3, std1=0.02,
                              explore_count=13, nominal_candidate=-1, param_variance=var)
    ctx.set_state([0, 0, 0, 0], 0.0, [0.25, 0, 0.01, 1, 0, 0, 0])
    ctx.rollout_noise(N, H, 0, times, nominal, ns)
    ref_nodes = pyoracle.noise_candidates(pm, ns, P, nominal, range(N))
    for i in (0, 12, 13, 100):
        assert np.allclose(ctx.fetch_spline(i), ref_nodes[i], rtol=0, atol=1e-13)


# ------------------------------------------------------------------ selection + layout
def test_topk_matches_sorted_returns(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 1000, 12, 3
    ctx.set_state([0, 0.5, 0, 0], 0.0)
    ctx.rollout_splines(H, 2, np.linspace(0, 0.11, P), random_nodes(8, N, P, 1))
    ret, _ = ctx.returns()
    order = np.lexsort((np.arange(N), ret))
    for k in (1, 2, 10, 137, N):
        idx, vals = ctx.topk(k)
        assert np.array_equal(idx, order[:k]) and np.array_equal(vals, ret[order[:k]])
    assert ctx.return_of(5) == ret[5]


def test_fused_best(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 777, 12, 4
    ctx.set_state([0, 0.5, 0, 0], 0.0)
    ctx.rollout_noise(N, H, 2, np.linspace(0, 0.11, P), np.zeros((P, 1)), capi.make_noise_spec(seed=11, std0=0.5))
    ret, _ = ctx.returns()
    idx, best, ref, spline = ctx.best(ref_candidate=0)
    assert idx == int(np.lexsort((np.arange(N), ret))[0]) and best == ret[idx] and ref == ret[0]
    assert np.array_equal(spline, ctx.fetch_spline(idx))
    idx2, _, ref2, sp2 = ctx.best(ref_candidate=-1, with_spline=False)
    assert idx2 == idx and np.isnan(ref2) and sp2 is None


def test_api_errors(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    with pytest.raises(capi.MjpcxError) as e:
        ctx.returns()
    assert e.value.code == -5                                   # MJPCX_ESTATE
    with pytest.raises(capi.MjpcxError) as e:
        ctx.N = 4
        ctx.rollout_splines(8, 2, [0.0, 0.0], np.zeros((4, 2, 1)))  # node times must increase
    assert e.value.code == -1
    ctx.rollout_splines(8, 2, [0.0, 0.1], np.zeros((4, 2, 1)))
    with pytest.raises(capi.MjpcxError):
        ctx.fetch_trajectory(4)
    with pytest.raises(capi.MjpcxError):
        ctx.topk(5)


# ------------------------------------------------------------------ BASELINE.json full-size configs
def test_config1_cartpole_n8_h64(cartpole):
    """configs[0]: Cartpole PS, 8 candidates, horizon 64 (the reference's CPU-runnable plumbing case)"""
    N, H, P = 8, 64, 10
    times = np.linspace(0, 0.63, P)
    pm = cartpole.packed_model()
    nodes = pyoracle.noise_candidates(pm, capi.make_noise_spec(seed=0, std0=0.5), P, np.zeros((P, 1)), range(N))
    compare_batch(cartpole, [1.0, 0.0, 0.0, 0.0], 0.0, None, N, H, P, 2, times, nodes)


def test_config2_cartpole_n4096_h128_properties(cartpole):
    """configs[1] at full size: 4096 candidates x horizon 128, fp64. Oracle on a sample + properties."""
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 4096, 128, 10
    times = np.linspace(0, 1.27, P)
    nominal = np.zeros((P, 1))
    state = [1.0, 0.0, 0.0, 0.0]                               # home keyframe
    ns = capi.make_noise_spec(seed=0, iteration=0, std0=0.5)
    ctx.set_state(state, 0.0)
    ctx.rollout_noise(N, H, 2, times, nominal, ns)
    ret, fail = ctx.returns()
    assert not fail.any() and np.all(np.isfinite(ret)) and np.all(ret > 0)
    # (a) determinism: the same launch twice is bit-identical
    ctx.rollout_noise(N, H, 2, times, nominal, ns)
    assert np.array_equal(ctx.returns()[0], ret)
    # (b) candidate 0 is the un-noised nominal: equals an explicit rollout of the nominal spline
    idx, best = ctx.topk(1)
    tr0 = ctx.fetch_trajectory(0)
    ref0 = pyoracle.rollout_batch(pm, pt, state, 0.0, None, 1, H, P, 2, times, nominal[None])
    assert close(tr0.states, ref0["states"][0]) and close(ret[0], ref0["total_return"][0])
    # (c) return == mean of costs, costs == CostValue(residual) for the winner
    trw = ctx.fetch_trajectory(int(idx[0]))
    assert abs(trw.total_return - trw.costs.mean()) < 1e-12 and trw.total_return == ret[idx[0]] == best[0]
    for k in (0, 50, 127):
        assert abs(trw.costs[k] - pyoracle.cost_value(pt, trw.residual[k])) < 1e-12
    # (d) oracle on a strided sample of 128 candidates
    sample = np.arange(0, N, 32)
    nodes = pyoracle.noise_candidates(pm, ns, P, nominal, sample)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, None, len(sample), H, P, 2, times, nodes, num_threads=8)
    assert close(ret[sample], ref["total_return"])
    assert ret[idx[0]] == ret.min()


def test_fp32_kernel(cartpole):
    N, H, P = 128, 32, 4
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 32)
    nodes, times = random_nodes(21, N, P, 1), np.linspace(0, 0.31, P)
    ctx.set_state([0.2, 2.9, 0.1, -0.2], 0.0)
    ctx.rollout_splines(H, 2, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, [0.2, 2.9, 0.1, -0.2], 0.0, None, N, H, P, 2, times, nodes)
    assert not fail.any() and close(ret, ref["total_return"], 2e-3)
    assert ctx.algorithmic_bytes(H, P) * 2 == capi.Context(pm, pt, 0, 64).algorithmic_bytes(H, P)


@pytest.mark.parametrize("name,state,mocap", [("Cartpole", [0.2, 2.9, 0.1, -0.2], None),
                                              ("Particle", [0.05, -0.1, 0.2, 0.0], [0.15, -0.1, 0.01, 1, 0, 0, 0])])
@pytest.mark.parametrize("precision,tol", [(64, 1e-9), (32, 2e-3)])
def test_noisy_rollout_on_the_lane_kernels(name, state, mocap, precision, tol):
    """Trajectory::NoisyRollout (trajectory.cc:147-155) on the lane-per-candidate family: Ornstein-Uhlenbeck xfrc_applied noise
    from the shared counter-based stream, accumulated through mj_xfrcAccumulate; each candidate tracks the oracle's rollout
    with the same noise, identical splines diverge, and xfrc_std = 0 is the plain rollout bit for bit."""
    task = load_task(name)
    pm, pt = task.packed_model(), task.packed()
    N, P, H = 70, 4, 40
    times = np.linspace(0, 0.39, P)
    nodes = np.tile(random_nodes(5, 1, P, task.model.nu), (N, 1, 1))
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 2, times, nodes, num_threads=4, xfrc_std=0.8, xfrc_rate=0.05,
                                 seed=3, candidate_offset=9)
    assert np.abs(ref["states"][0] - ref["states"][1]).max() > 1e-4
    ctx = capi.Context(pm, pt, 0, precision)
    assert "rollout_lane" in ctx.kernel_name
    ctx.set_state(state, 0.0, mocap)
    ctx.rollout_splines_noisy(H, 2, times, nodes, 0.8, 0.05, seed=3, candidate_offset=9)
    ret, fail = ctx.returns()
    assert np.array_equal(fail, ref["failure"]) and close(ret, ref["total_return"], tol)
    for c in (0, 37, N - 1):
        tr = ctx.fetch_trajectory(c)
        assert close(tr.states, ref["states"][c], tol) and close(tr.residual, ref["residual"][c], tol)
    ctx.rollout_splines_noisy(H, 2, times, nodes, 0.0, 0.05)
    r0, _ = ctx.returns()
    ctx.rollout_splines(H, 2, times, nodes)
    assert np.array_equal(r0, ctx.returns()[0])
    ctx.close()


@pytest.mark.parametrize("N,H,interp", [(70, 48, 2), (257, 16, 1), (33, 1, 0)])
def test_rk4_integrator_on_the_lane_kernels(cartpole, particle, N, H, interp):
    """mjINT_RK4 (agent_integrator / <option integrator="RK4">): mj_RungeKutta(4) inside every mj_step of the rollout, on the
    candidate-per-lane kernels, against oracle/physics.c o_rk4"""
    P = 4
    pm = cartpole.packed_model(); pm.struct.integrator = 1
    times = 0.25 + np.arange(P) * 0.01 * max(H - 1, 1) / (P - 1)
    ret, ref = compare_batch(cartpole, [0.3, 2.7, -0.4, 0.9], 0.25, None, N, H, P, interp, times, random_nodes(N + H, N, P, 1), pm=pm)
    if H > 1:  # and it IS a different integrator: the Euler rollout of the same splines differs
        eu = pyoracle.rollout_batch(cartpole.packed_model(), cartpole.packed(), [0.3, 2.7, -0.4, 0.9], 0.25, None, N, H, P, interp, times,
                                    random_nodes(N + H, N, P, 1), num_threads=8)
        assert np.abs(eu["total_return"] - ref["total_return"]).max() > 1e-6
    pm = particle.packed_model(); pm.struct.integrator = 1
    mocap = [0.2, -0.1, 0.01, 1, 0, 0, 0]
    compare_batch(particle, [0.05, -0.1, 0.3, 0.2], 0.0, mocap, N, H, P, interp, np.arange(P) * 0.1 * max(H - 1, 1) / (P - 1) , random_nodes(N, N, P, 2), pm=pm)


def test_implicitfast_is_the_euler_update_on_damping_only_models(cartpole):
    """<option integrator="implicitfast"> (mjpc tasks that ask for it through agent_integrator): for a model whose only velocity-dependent
    smooth force is joint damping, MuJoCo's M - h dqfrc_smooth/dqvel is M + h diag(damping) -- mj_implicit's update is mj_Euler's. The
    context accepts such a model and its rollouts are bit-identical to the Euler context's; with a velocity term in an actuator's bias
    (position actuator with kv) it is refused, as plain `implicit` is."""
    N, H, P = 64, 32, 4
    times = np.arange(P) * 0.01 * (H - 1) / (P - 1)
    nodes = random_nodes(3, N, P, 1)
    state = [0.3, 2.7, -0.4, 0.9]
    rets = []
    for integ in (0, 3):
        pm = cartpole.packed_model(); pm.struct.integrator = integ
        ctx = capi.Context(pm, cartpole.packed(), 0, 64)
        ctx.set_state(np.asarray(state, float), 0.0)
        ctx.rollout_splines(H, 2, times, nodes)
        rets.append(ctx.returns()[0].copy())
        ctx.close()
        ref = pyoracle.rollout_batch(pm, cartpole.packed(), state, 0.0, None, N, H, P, 2, times, nodes, num_threads=4)
        assert close(rets[-1], ref["total_return"], 1e-9)
    assert np.array_equal(rets[0], rets[1])
    pm = cartpole.packed_model(); pm.struct.integrator = 2
    with pytest.raises(capi.MjpcxError):
        capi.Context(pm, cartpole.packed(), 0, 64)
    pm = cartpole.packed_model(); pm.struct.integrator = 3
    bias = np.ctypeslib.as_array(pm.struct.actuator_biasprm, (3 * cartpole.model.nu,))
    btype = np.ctypeslib.as_array(pm.struct.actuator_biastype, (cartpole.model.nu,))
    saved = (float(bias[2]), int(btype[0]))  # (the packed arrays are the fixture model's own: put them back)
    try:
        bias[2] = -0.5
        if btype[0] != 1:  # (biastype none: biasprm is not read by MuJoCo, so the velocity coefficient does not count)
            capi.Context(pm, cartpole.packed(), 0, 64).close()
            btype[0] = 1   # affine
        with pytest.raises(capi.MjpcxError) as err:
            capi.Context(pm, cartpole.packed(), 0, 64)
        assert "velocity-dependent actuator" in str(err.value)
    finally:
        bias[2], btype[0] = saved
    # mj_implicit ignores mjDSBL_EULERDAMP, mj_Euler honours it: with the flag set the two updates differ and the model is refused
    pm = cartpole.packed_model(); pm.struct.integrator = 3
    pm.struct.disableflags |= 1 << 14
    with pytest.raises(capi.MjpcxError) as err:
        capi.Context(pm, cartpole.packed(), 0, 64)
    assert "eulerdamp" in str(err.value)
    assert pyoracle.lib().odata_new(pm.ptr) is None  # (the oracle refuses it as well)
