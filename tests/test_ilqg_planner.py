"""iLQG planner host logic (planners.py::GpuILQGPlanner): port of mjpc/test/ilqg_planner/ilqg_test.cc on the
oracle-backed backend (CPU) and on the GPU, plus GPU-vs-oracle-backend agreement."""
import numpy as np
import pytest

from mujoco_mpc_amd.planners import GpuILQGPlanner, State, log_scale, find_interval
from oracle_backend import OracleContext


def run_particle(factory, iterations=25, rollouts=10):
    from mujoco_mpc_amd.task import load_task
    task = load_task("Particle")
    p = GpuILQGPlanner(backend_factory=factory)
    p.initialize(task.model, task)
    p.num_rollouts_gui_ = rollouts
    p.allocate()
    p.reset(512)
    steps = int(max(min(2.5 / 0.1 + 1, 512), 1))          # ilqg_test.cc:75-79: horizon 2.5 s, timestep 0.1
    st = State(task.model)
    st.set([0.0, 0.0], [0.0, 0.0])
    p.set_state(st)
    hist = []
    for _ in range(iterations):
        p.optimize_policy(steps)
        hist.append((p.candidate0.trajectory.total_return, p.winner, p.regularization, p.action_step))
    return p, st, steps, hist


def check_converged(p, st, steps):
    tr = p.candidate0.trajectory
    assert abs(tr.states[steps - 1, 0] - st.mocap[0]) < 1e-2      # ilqg_test.cc:96-103
    assert abs(tr.states[steps - 1, 1] - st.mocap[1]) < 1e-2
    assert abs(tr.states[steps - 1, 2]) < 1e-1 and abs(tr.states[steps - 1, 3]) < 1e-1
    assert np.all(np.abs(tr.actions[:steps - 1]) <= 1.0)           # :108-119


def test_helpers():
    s = log_scale(1.0, 1e-3, 9)
    assert abs(s[0] - 1e-3) < 1e-15 and abs(s[-1] - 1.0) < 1e-12 and np.all(np.diff(s) > 0)
    xs = [0.0, 1.0, 2.0, 3.0]
    assert find_interval(xs, -1.0, 4) == (0, 0) and find_interval(xs, 0.5, 4) == (0, 1)
    assert find_interval(xs, 3.0, 4) == (3, 3) and find_interval(xs, 2.0, 3) == (2, 2)   # agent_utilities_test.cc
    assert GpuILQGPlanner.best_rollout([3.0, 1.0, 1.0, 2.0], [0, 0, 0, 0]) == 2          # ties: last index scanned first
    assert GpuILQGPlanner.best_rollout([3.0, 1.0], [1, 1]) == -1


def test_ilqg_particle_oracle_backend():
    p, st, steps, hist = run_particle(lambda t: OracleContext(t, differentiable=True), iterations=25)
    check_converged(p, st, steps)
    assert hist[-1][0] < hist[0][0]
    a = np.zeros(2)
    p.action_from_policy(a, np.zeros(4), 0.05)
    assert np.all(np.abs(a) <= 1.0)


@pytest.mark.gpu
def test_ilqg_particle_gpu():
    p, st, steps, hist = run_particle(None, iterations=25)
    check_converged(p, st, steps)
    # a finer line search than the reference's 10 rollouts costs nothing on the device
    p2, st2, steps2, hist2 = run_particle(None, iterations=25, rollouts=256)
    check_converged(p2, st2, steps2)
    assert hist2[-1][0] <= hist[-1][0] * 1.05


@pytest.mark.gpu
def test_ilqg_gpu_tracks_oracle_backend():
    g, _, steps, hg = run_particle(None, iterations=6)
    o, _, _, ho = run_particle(lambda t: OracleContext(t, differentiable=True), iterations=6)
    for (rg, wg, mg, sg), (ro, wo, mo, so) in zip(hg, ho):
        assert wg == wo and mg == mo and sg == so
        assert abs(rg - ro) < 1e-7 * (1 + abs(ro))
    assert np.allclose(g.policy.feedback_gain[:steps], o.policy.feedback_gain[:steps], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("representation", [0, 1, 2])
def test_oracle_time_policy_matches_the_python_policy(representation):
    """iLQGPolicy::Action (ilqg/policy.cc:82-161) in its three representations -- zero-order, linear, cubic (Hermite with
    finite-difference slopes, utilities.cc:336-422): the C oracle's time-based feedback rollout against a step-by-step rollout of
    the oracle's physics driven by the Python ILQGPolicy (two implementations of the interpolation)."""
    from mujoco_mpc_amd.planners import ILQGPolicy
    from mujoco_mpc_amd.task import load_task
    from oracle import pyoracle
    task = load_task("Particle")
    pm, pt = task.packed_model(), task.packed()
    m = task.model
    H = 12
    rng = np.random.default_rng(3)
    pol = ILQGPolicy(m, task)
    pol.reset(H)
    pol.representation = representation
    tr = pol.trajectory
    tr.horizon = H
    dt = float(pm.struct.timestep)  # the planning copy of the model steps at agent_timestep
    tr.times[:H] = 2.3 * dt + dt * np.arange(H)  # knots off the step grid: every step interpolates
    tr.states[:H] = 0.1 * rng.normal(size=(H, 4))
    tr.actions[:H] = 0.3 * rng.normal(size=(H, 2))
    pol.feedback_gain[:H] = 0.4 * rng.normal(size=(H, 2, 4))
    pol.feedback_scaling = 0.7
    state0 = np.array([0.05, -0.02, 0.1, 0.0])
    t0 = 0.0  # starts BEFORE the first knot: the policy extrapolates by holding the first value
    ref = pyoracle.rollout_feedback(pm, pt, state0, t0, np.zeros(7), H + 4, 1, representation, 1, tr.times[:H], tr.states[:H], tr.actions[:H],
                                    pol.feedback_gain[:H], np.zeros((H, 2)), np.array([pol.feedback_scaling]))
    ph = pyoracle.Physics(pm)
    x, t = state0.copy(), t0
    for k in range(H + 3):
        u = np.zeros(2)
        pol.action(u, x, t)
        assert np.allclose(ref["actions"][0][k], u, rtol=0, atol=1e-12), (representation, k)
        ph.set_state(x[:2], x[2:], t, np.zeros(7))
        ph.set_ctrl(u)
        ph.step()
        x = np.concatenate([ph.get("qpos"), ph.get("qvel")])
        t += dt
        assert np.allclose(ref["states"][0][k + 1], x, rtol=0, atol=1e-12)


def test_oracle_ilqg_sweeps_do_not_depend_on_the_thread_count():
    """oracle/ilqg.c fans the derivative sweep over time steps and the feedback rollouts over candidates (bench.py's CPU leg of
    configs[4] uses 16 workers, as the reference's ThreadPool would); every item has its own physics arena, so the numbers are
    the single-thread ones bit for bit. On the A1: free joint, contacts."""
    from mujoco_mpc_amd.task import load_task
    from oracle import pyoracle
    t = load_task("QuadrupedFlat")
    t.transition(0.0)
    pm, pt = t.packed_model(), t.packed()
    rng = np.random.default_rng(4)
    H = 10
    state = np.concatenate([t.model.keyframes["home"]["qpos"], 0.1 * rng.normal(size=18)])
    mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
    times = np.arange(3) * (H - 1) * 0.01 / 2
    nodes = np.clip(rng.normal(0, 0.1, (1, 3, 12)), -1, 1)
    nom = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, 1, H, 3, 1, times, nodes, num_threads=1)
    nom = {k: v[0] for k, v in nom.items() if k not in ("total_return", "failure")}
    one = pyoracle.transition_fd(pm, pt, nom["states"], nom["times"], nom["actions"], 1e-6, 0, mocap=mocap, num_threads=1)
    four = pyoracle.transition_fd(pm, pt, nom["states"], nom["times"], nom["actions"], 1e-6, 0, mocap=mocap, num_threads=4)
    assert all(np.array_equal(a, b) for a, b in zip(one, four)) and np.abs(one[0]).max() > 0.5
    gains = 0.05 * rng.normal(size=(H, 12, 36))
    improvement = 0.05 * rng.normal(size=(H, 12))
    alpha = np.exp(np.linspace(0, np.log(1e-3), 7))
    args = (pm, pt, state, 0.0, mocap, H, 0, 0, 1, nom["times"], nom["states"], nom["actions"], gains, improvement, alpha)
    a, b = pyoracle.rollout_feedback(*args, num_threads=1), pyoracle.rollout_feedback(*args, num_threads=3)
    assert all(np.array_equal(a[k], b[k]) for k in a) and np.ptp(a["total_return"]) > 0
