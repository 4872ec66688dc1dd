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
