"""-m gpu: the fp32 instantiation of the wavefront-per-candidate kernel (precision = 32; BASELINE configs[3] is quoted
in fp32) against the fp64 oracle. Stated tolerance: total returns within 1e-4 relative over 40 steps (observed ~1e-5: the
physics is the same arithmetic in float, the Newton solver stops at the float noise floor) and within 3e-2 over 100 steps
(contact-rich legged dynamics amplify the rounding: observed 1e-2 on the A1, 3e-6 on the falling humanoid), states within
3e-2 over 40 steps (observed 2e-2 on joint velocities of ~3 rad/s of the A1's flailing legs with the Jacobian-free
constraint path, 2e-3 on the humanoid; the returns agree to 1e-5); candidates whose fp64 rollout fails are excluded."""
import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def mocap7(mpos):
    return np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(mpos).reshape(-1, 3)])


def setup(name):
    t = load_task(name)
    if name == "QuadrupedFlat":
        t.transition(0.0)
        return t, t.model.keyframes["home"]["qpos"], np.zeros(18), np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0]), 0.1
    e = t.transition(0.0, mode=9)
    return t, e["qpos"], e["qvel"], mocap7(e["mocap_pos"]), 0.3


@pytest.mark.parametrize("name", ["QuadrupedFlat", "HumanoidTrack"])
@pytest.mark.parametrize("H,rtol,stol", [(5, 1e-4, 2e-4), (40, 1e-4, 3e-2), (100, 3e-2, None)])
def test_fp32_rollouts_track_the_fp64_oracle(name, H, rtol, stol):
    t, q, v, mocap, std = setup(name)
    pm, pt = t.packed_model(), t.packed()
    N, P = 16, 4
    rng = np.random.default_rng(H)
    dt = t.model.get_number("agent_timestep", t.model.timestep)
    times = np.arange(P) * max((H - 1) * dt / (P - 1), 1e-3)
    nodes = np.clip(rng.normal(0, std, (N, P, t.model.nu)), -1, 1)
    state = np.concatenate([q, v])
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 1, times, nodes, num_threads=8)
    ctx = capi.Context(pm, pt, 0, 32)
    assert "rollout_wave_kernel" in ctx.kernel_name or "rollout_tree_kernel" in ctx.kernel_name
    ctx.set_state(state, 0.0, mocap)
    ctx.rollout_splines(H, 1, times, nodes)
    ret, fail = ctx.returns()
    ok = ref["failure"] == 0
    assert ok.sum() >= N // 2 and not fail[ok].any()
    rel = np.abs(ret[ok] - ref["total_return"][ok]) / np.abs(ref["total_return"][ok])
    assert rel.max() < rtol, rel.max()
    if stol is not None:
        c = int(np.flatnonzero(ok)[0])
        tr = ctx.fetch_trajectory(c)
        assert np.abs(tr.states - ref["states"][c]).max() < stol
        assert np.abs(tr.residual - ref["residual"][c]).max() < 50 * stol
    ctx.close()


@pytest.mark.parametrize("name", ["QuadrupedFlat", "HumanoidTrack"])
def test_fp32_rk4_rollouts_track_the_fp64_oracle(name):
    """mjINT_RK4 in the float instantiation (w32::rollout_wave_kernel<32, false, true>): the A1 (elliptic cones, friction loss) and
    the humanoid (pyramidal cones, tendons) over 20 steps = 80 forward passes"""
    t, q, v, mocap, std = setup(name)
    pm, pt = t.packed_model(), t.packed()
    pm.struct.integrator = 1
    N, P, H = 8, 3, 20
    rng = np.random.default_rng(3)
    dt = t.model.get_number("agent_timestep", t.model.timestep)
    times = np.arange(P) * (H - 1) * dt / (P - 1)
    nodes = np.clip(rng.normal(0, std, (N, P, t.model.nu)), -1, 1)
    state = np.concatenate([q, v])
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 1, times, nodes, num_threads=8)
    eul = pyoracle.rollout_batch(t.packed_model(), pt, state, 0.0, mocap, N, H, P, 1, times, nodes, num_threads=8)
    assert not ref["failure"].any()
    ctx = capi.Context(pm, pt, 0, 32)
    ctx.set_state(state, 0.0, mocap)
    ctx.rollout_splines(H, 1, times, nodes)
    ret, fail = ctx.returns()
    assert not fail.any()
    rel = np.abs(ret - ref["total_return"]) / np.abs(ref["total_return"])
    assert rel.max() < 1e-4, rel.max()
    # ... and closer to the RK4 oracle than the Euler rollout of the same splines is (it IS the other integrator)
    assert np.abs(ret - ref["total_return"]).max() < 0.1 * np.abs(eul["total_return"] - ref["total_return"]).max()
    tr = ctx.fetch_trajectory(0)
    assert np.abs(tr.states - ref["states"][0]).max() < 2e-2
    ctx.close()


def test_fp32_planner_on_the_humanoid():
    """the C++ Predictive-Sampling planner at precision 32 improves the tracking cost like the fp64 one"""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    t = load_task("HumanoidTrack")
    m = t.model
    scores = {}
    for prec in (64, 32):
        p = HostPlanner(t, seed=5, num_trajectory=128, kind="sampling", precision=prec)
        q, v, mp = np.array(m.qpos0, float), np.zeros(27), np.zeros(48)
        p.task_transition_state(0.0, 9, q, v, mp)
        p.reset(32)
        mq = np.tile([1.0, 0, 0, 0], (16, 1))
        s = []
        for k in range(3):
            p.set_state(q, v, 0.0, mocap_pos=mp.reshape(16, 3), mocap_quat=mq)
            p.optimize_policy(32)
            s.append(p.best_score)
        scores[prec] = s
    assert scores[32][-1] <= scores[32][0]
    assert abs(scores[32][0] - scores[64][0]) / scores[64][0] < 1e-3   # same noise, same winner up to float rounding
