"""-m gpu: the GPU-backed sampling planner end to end (port of
mjpc/test/sampling_planner/sampling_planner_test.cc) and against the oracle-backed planner."""
import numpy as np
import pytest

from mujoco_mpc_amd.planners import GpuSamplingPlanner, State
from oracle_backend import OracleContext

pytestmark = pytest.mark.gpu


def test_gpu_planner_converges_on_particle(particle):
    task = particle
    p = GpuSamplingPlanner(device=0, seed=1)
    p.initialize(task.model, task)
    p.num_trajectory_ = 256          # above the reference's kMaxTrajectory = 128
    p.noise_exploration[0] = 0.1
    p.allocate()
    H = task.planning_steps()
    p.reset(H)
    st = State(task.model); st.set([0.0, 0.0], [0.0, 0.0])
    p.set_state(st)
    for _ in range(60):
        p.optimize_policy(H)
    best = p.best_trajectory()
    assert np.abs(best.states[-1, :2] - st.mocap[:2]).max() < 0.1
    assert np.all(np.abs(best.actions) <= 1.0)
    assert p.ctx.kernel_name.startswith("rollout_lane<TopoParticle,TaskParticle")


def test_gpu_planner_tracks_oracle_planner(cartpole):
    """same seeds -> the GPU planner and the oracle-backed planner pick the same winners"""
    def make(factory):
        p = GpuSamplingPlanner(device=0, seed=9, backend_factory=factory)
        p.initialize(cartpole.model, cartpole); p.num_trajectory_ = 200; p.allocate()
        p.reset(32)
        st = State(cartpole.model); st.set([0.1, 2.6], [0.0, 0.2])
        p.set_state(st)
        return p
    g, o = make(None), make(lambda t: OracleContext(t, threads=4))
    for _ in range(4):
        g.optimize_policy(32); o.optimize_policy(32)
        assert g.winner == o.winner
        assert abs(g.candidate_score(0) - o.candidate_score(0)) < 1e-9 * (1 + abs(o.candidate_score(0)))
        assert np.allclose(g.policy.plan.values(), o.policy.plan.values(), rtol=0, atol=1e-12)
    gt, ot = g.best_trajectory(), o.best_trajectory()
    assert np.allclose(gt.states, ot.states, rtol=0, atol=1e-9)
