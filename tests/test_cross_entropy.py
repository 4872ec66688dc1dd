"""Cross-Entropy planner (mjpc/planners/cross_entropy/planner.cc) host logic on the oracle-backed
backend, and against an independent numpy restatement of the elite update (planner.cc:216-283)."""
import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.planners import GpuCrossEntropyPlanner, State
from oracle import pyoracle
from oracle_backend import OracleContext


def make(task, n, factory=lambda t: OracleContext(t), **kw):
    p = GpuCrossEntropyPlanner(backend_factory=factory, **kw)
    p.initialize(task.model, task)
    p.num_trajectory_ = n
    p.n_elite_ = max(n // 10, 2)
    p.allocate()
    return p


def test_initialize_defaults(particle):
    p = GpuCrossEntropyPlanner(backend_factory=lambda t: OracleContext(t))
    p.initialize(particle.model, particle)
    assert p.std_initial_ == 0.01 and p.std_min_ == 0.01 and p.explore_fraction_ == 0.0
    assert p.num_trajectory_ == 10 and p.n_elite_ == 2         # max(N/10, 2), planner.cc:68-69
    assert p.interpolation_ == 0                                # kZeroSpline member default


def test_one_iteration_matches_numpy_restatement(cartpole):
    task = cartpole
    N, H = 40, 30
    p = make(task, N, seed=4)
    p.std_initial_, p.std_min_, p.explore_fraction_ = 0.3, 0.05, 0.25
    p.reset(H)
    st = State(task.model); st.set([0.1, 0.4], [0.0, 0.0], time=0.2)
    p.set_state(st)
    p.optimize_policy(H)
    # ---- independent restatement
    pm, pt = task.packed_model(), task.packed()
    P = 10
    times = 0.2 + np.arange(P) * max((H - 1) * 0.01 / (P - 1), 1e-5)
    assert np.allclose(p.policy.plan.times(), times, rtol=0, atol=1e-15)
    var0 = np.full(P, 0.3 ** 2)
    ns = capi.make_noise_spec(seed=4, iteration=0, mode=capi.NOISE_CROSS_ENTROPY, nominal_candidate=N,
                              explore_count=10, std0=0.3, std1=0.05, param_variance=var0)
    nodes = pyoracle.noise_candidates(pm, ns, P, np.zeros((P, 1)), range(N))
    ref = pyoracle.rollout_batch(pm, pt, st.state, 0.2, None, N, H, P, 0, times, nodes)
    order = np.lexsort((np.arange(N), ref["total_return"]))[:4]
    mean = nodes[order].mean(axis=0)
    var = ((nodes[order] - mean) ** 2).sum(axis=0) / (4 - 1)
    assert p.trajectory_order == [int(i) for i in order]
    assert np.allclose(p.policy.plan.values(), mean, rtol=0, atol=1e-15)
    assert np.allclose(p.variance[:P], var.reshape(-1), rtol=1e-13, atol=1e-18)
    assert abs(p.improvement - max(ref["total_return"][order].mean() - ref["total_return"][order[0]], 0)) < 1e-14
    # BestTrajectory is the NOMINAL rollout (planner.cc:446-448)
    nom = pyoracle.rollout_batch(pm, pt, st.state, 0.2, None, 1, H, P, 0, times, np.zeros((1, P, 1)))
    assert np.array_equal(p.best_trajectory().states, nom["states"][0])


def test_cross_entropy_converges_on_particle(particle):
    p = make(particle, 64, seed=2)
    p.std_initial_, p.std_min_ = 0.3, 0.02
    H = particle.planning_steps()
    p.reset(H)
    st = State(particle.model); st.set([0.0, 0.0], [0.0, 0.0])
    p.set_state(st)
    for _ in range(25):
        p.optimize_policy(H)
    best = p.nominal_trajectory(H)
    assert np.abs(best.states[-1, :2] - st.mocap[:2]).max() < 0.1
    a = np.zeros(2)
    p.action_from_policy(a, None, 0.3)
    assert np.all(np.abs(a) <= 1.0)


@pytest.mark.gpu
def test_gpu_cross_entropy_tracks_oracle_backend(particle):
    def run(factory):
        p = make(particle, 96, factory=factory, seed=6) if factory else None
        if p is None:
            p = GpuCrossEntropyPlanner(device=0, seed=6)
            p.initialize(particle.model, particle); p.num_trajectory_ = 96; p.n_elite_ = 9; p.allocate()
        p.std_initial_, p.std_min_, p.explore_fraction_ = 0.2, 0.02, 0.1
        p.reset(11)
        st = State(particle.model); st.set([0.05, -0.1], [0.0, 0.1])
        p.set_state(st)
        out = []
        for _ in range(3):
            p.optimize_policy(11)
            out.append((list(p.trajectory_order), p.policy.plan.values().copy(), p.variance[:22].copy(), p.improvement))
        return out, p.best_trajectory()
    (g, gb), (o, ob) = run(None), run(lambda t: OracleContext(t))
    for (go, gv, gvar, gi), (oo, ov, ovar, oi) in zip(g, o):
        assert go == oo
        assert np.allclose(gv, ov, rtol=0, atol=1e-12) and np.allclose(gvar, ovar, rtol=1e-9, atol=1e-15)
        assert abs(gi - oi) < 1e-10
    assert np.allclose(gb.states, ob.states, rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_elite_moments_on_device(cartpole):
    pm, pt = cartpole.packed_model(), cartpole.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    N, H, P = 500, 8, 6
    ctx.set_state([0, 0.2, 0, 0], 0.0)
    ctx.rollout_noise(N, H, 1, np.linspace(0, 0.07, P), np.zeros((P, 1)), capi.make_noise_spec(seed=2, std0=0.5))
    ret, _ = ctx.returns()
    nodes = np.stack([ctx.fetch_spline(i) for i in range(0, N, 7)])
    cand = np.arange(0, N, 7)
    s, sr = ctx.elite_moments(cand)
    assert np.allclose(s, nodes.sum(axis=0), rtol=1e-13) and abs(sr - ret[cand].sum()) < 1e-10
    mean = s / len(cand)
    sq, _ = ctx.elite_moments(cand, mean)
    assert np.allclose(sq, ((nodes - mean) ** 2).sum(axis=0), rtol=1e-12)
    s0, sr0 = ctx.elite_moments([])
    assert np.all(s0 == 0) and sr0 == 0
