"""-m gpu: the wavefront-per-candidate kernel (since round 3: the registered Jacobian-free one, rollout_tree_kernel<Humanoid>) on the Humanoid mocap-tracking task of BASELINE configs[3]
(28/27/21, 37 bodies, pyramidal cones, capsule self-collision, fixed-tendon limits, the 141-entry tracking residual)
against the CPU oracle. Tolerances as for the Quadruped: 1e-9 (1 + |x|) over the first steps, looser with the horizon."""
import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def tree_kernel_only(monkeypatch):
    """this suite is the wavefront-per-candidate kernel's (the hand-on target of the limb kernel, the kernel of small batches and NoisyRollout):
    contexts are created without the limb kernel, whose own suite is tests/test_gpu_limb.py"""
    monkeypatch.setenv("MJPCX_NO_LIMB", "1")


def mocap7(mpos):
    return np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(mpos).reshape(-1, 3)])


@pytest.fixture(scope="module")
def walk():
    t = load_task("HumanoidTrack")
    e = t.transition(0.0, mode=9)   # Walk
    return t, e


def close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= tol * (1 + np.abs(b)))


def run(task, state, mocap, N, H, P, interp, seed, tol, time=0.0, std=0.3):
    pm, pt = task.packed_model(), task.packed()
    rng = np.random.default_rng(seed)
    dt = task.model.get_number("agent_timestep", task.model.timestep)
    times = time + np.arange(P) * max((H - 1) * dt / max(P - 1, 1), 1e-3)
    nodes = np.clip(rng.normal(0, std, (N, P, task.model.nu)), -1, 1)
    ctx = capi.Context(pm, pt, 0, 64)
    assert "rollout_tree_kernel<Humanoid>" in ctx.kernel_name
    ctx.set_state(state, time, mocap)
    ctx.rollout_splines(H, interp, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, time, mocap, N, H, P, interp, times, nodes, num_threads=8)
    assert np.array_equal(fail, ref["failure"]) and not fail.any()
    for c in range(N):
        tr = ctx.fetch_trajectory(c)
        for name in ("states", "actions", "times", "residual", "costs", "trace"):
            g, o = getattr(tr, name), ref[name][c]
            assert close(g, o, tol), (name, c, float(np.max(np.abs(g - o))))
    assert close(ret, ref["total_return"], tol)
    ctx.close()


def test_walk_first_steps(walk):
    t, e = walk
    run(t, np.concatenate([e["qpos"], e["qvel"]]), mocap7(e["mocap_pos"]), N=4, H=5, P=4, interp=2, seed=1, tol=1e-9)


@pytest.mark.parametrize("interp", [0, 2])
def test_walk_forty_steps(walk, interp):
    t, e = walk
    run(t, np.concatenate([e["qpos"], e["qvel"]]), mocap7(e["mocap_pos"]), N=6, H=40, P=8, interp=interp, seed=3 + interp, tol=1e-6)


def test_collapse_exercises_self_collision_and_tendons():
    """zero-ish controls from a crouch: the body folds, arms and legs touch (capsule-capsule, sphere-capsule pairs), the
    hamstring tendons reach their limits and many pyramid rows are active"""
    t = load_task("HumanoidTrack")
    e = t.transition(0.0, mode=4)   # Crouch Flip
    q = e["qpos"].copy()
    v = np.zeros(27)
    v[3:6] = [1.5, -1.0, 0.5]
    run(t, np.concatenate([q, v]), mocap7(e["mocap_pos"]), N=4, H=80, P=4, interp=0, seed=9, tol=1e-5, std=0.8)


def test_later_frames_of_the_motion(walk):
    """time inside the clip: the residual interpolates between two keyframes"""
    t = load_task("HumanoidTrack")
    t.transition(0.0, mode=8)          # Run
    e = t.transition(0.4133, mode=8)   # between keys
    q = np.array(t.model.arrays["key_qpos"][t.motion_start(8)], float)
    run(t, np.concatenate([q, np.zeros(27)]), mocap7(e["mocap_pos"]), N=3, H=12, P=4, interp=1, seed=11, tol=1e-8, time=0.4133)


def test_contact_feature_scene():
    """tests/models/capsules_tendon.xml on the device: pyramidal sphere-plane contact, crossed and exactly parallel free
    capsules (the two-contact branch), a limited fixed tendon; null residual (the cost is a dummy term)."""
    import os
    from mujoco_mpc_amd import mjcf
    from mujoco_mpc_amd.task import Task
    fm = mjcf.load_xml(os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "capsules_tendon.xml"))
    task = Task(name="scene", residual_id=0, model=fm).reset()
    q = fm.arrays["qpos0"].copy()
    v = np.zeros(fm.nv)
    q[14 + 2] = 1.0 + 0.0995      # rod_b just into rod_a
    v[12 + 2] = -0.5
    q[fm.nq - 2] = 0.6            # tendon beyond its upper limit
    v[0] = 0.3                    # the puck slides/rolls
    pm, pt = task.packed_model(), task.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    assert "rollout_wave_kernel" in ctx.kernel_name   # (an unregistered model: the generic kernel, Jacobian-free path, two passes)
    state = np.concatenate([q, v])
    H, P, N = 60, 2, 2
    times = np.array([0.0, 1.0])
    nodes = np.zeros((N, P, 1))
    nodes[1] = 0.5
    ctx.set_state(state, 0.0, np.zeros(0))
    ctx.rollout_splines(H, 0, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, np.zeros(0), N, H, P, 0, times, nodes, num_threads=2)
    assert not fail.any() and not ref["failure"].any()
    tr = ctx.fetch_trajectory(1)
    assert close(tr.states, ref["states"][1], 1e-7), float(np.max(np.abs(tr.states - ref["states"][1])))
    # something happened: the rods exchanged momentum and the tendon pushed the arm back
    assert tr.states[-1, fm.nq + 6 + 2] < -0.05 and tr.states[-1, fm.nq - 2] < 0.6
    ctx.close()


def test_cpp_predictive_sampling_on_the_humanoid_equals_python_planner():
    """BASELINE configs[3] in miniature: the C++ GpuSamplingPlanner (16 cubic spline points) on the tracking task, the
    C++ humanoid::Tracking transition providing the initial state, against the Python mirror with the same seed."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuSamplingPlanner, State
    t = load_task("HumanoidTrack")
    m = t.model
    N, H = 64, 32
    cpp = HostPlanner(t, seed=5, num_trajectory=N, kind="sampling")
    q, v, mp = np.array(m.qpos0, float), np.zeros(27), np.zeros(48)
    cpp.task_transition_state(0.0, 9, q, v, mp)
    e = t.transition(0.0, mode=9)
    assert np.array_equal(q, e["qpos"]) and np.array_equal(v, e["qvel"]) and np.array_equal(mp.reshape(16, 3), e["mocap_pos"])
    cpp.reset(H)
    assert cpp.num_spline_points == 16
    py = GpuSamplingPlanner(seed=5)
    py.initialize(m, t); py.num_trajectory_ = N; py.allocate(); py.reset(H)
    st = State(m)
    mq = np.tile([1.0, 0, 0, 0], (16, 1))
    scores = []
    for k in range(3):
        tm = 0.005 * k
        st.set(q, v, mocap_pos=mp.reshape(16, 3), mocap_quat=mq, time=tm); py.set_state(st); py.optimize_policy(H)
        cpp.set_state(q, v, tm, mocap_pos=mp.reshape(16, 3), mocap_quat=mq); cpp.optimize_policy(H)
        assert cpp.winner == py.winner and cpp.best_score == py.candidate_score(0)
        ct, cv = cpp.policy()
        assert np.array_equal(cv, py.policy.plan.values())
        scores.append(cpp.best_score)
    assert "rollout_tree_kernel<Humanoid>" in cpp.kernel_name and scores[-1] <= scores[0]


def test_a_folded_body_with_more_contacts_than_the_first_pass_stages():
    """a folded humanoid pressed into the floor: 36 contacts / 135 constraint rows in the oracle (MuJoCo-sized arena) -- more than the
    row-table kernel of rounds 1-2 could stage (16 contacts / 64 rows: such a candidate came back failed) and more than the first pass
    of the Jacobian-free path keeps in LDS (16 cones). The candidate overflows the short lists, is flagged, rolled out again by the second
    launch with the long lists, and comes back like the oracle's: no failure, same return."""
    t = load_task("HumanoidTrack")
    e = t.transition(0.0, mode=0)
    q = np.array([-0.04934, -0.00198, 0.032594, 0.351929, -0.166098, 0.001753, 0.92117, -0.571485, 0.278631, -0.418609, 0.178522, 0.04022,
                  0.050714, 0.63989, -0.043285, 0.662039, 0.162139, 0.387659, -0.046413, 0.214907, 0.638488, 0.143949, 0.316314, -0.489671,
                  -0.674248, 0.724606, -0.506946, -0.298564])
    q[3:7] /= np.linalg.norm(q[3:7])
    pm, pt = t.packed_model(), t.packed()
    mocap = mocap7(e["mocap_pos"])
    ph = pyoracle.Physics(pm)
    ph.set_state(q, np.zeros(27), 0.0, mocap)
    ph.set_ctrl(np.zeros(21))
    ph.forward_task(pt)
    assert int(ph.get("ncon")[0]) > 16 and int(ph.get("nefc")[0]) > 64
    N, H, P = 2, 4, 2
    dt = t.model.get_number("agent_timestep", t.model.timestep)
    times = np.arange(P) * max((H - 1) * dt, 1e-3)
    nodes = np.zeros((N, P, t.model.nu))
    state = np.concatenate([q, np.zeros(27)])
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.0, mocap)
    ctx.rollout_splines(H, 0, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 0, times, nodes, num_threads=2)
    assert not ref["failure"].any() and not fail.any()
    assert close(ret, ref["total_return"], 1e-6)
    for c in range(N):
        tr = ctx.fetch_trajectory(c)
        assert close(tr.states, ref["states"][c], 1e-6)
    ctx.close()


def test_random_states_against_the_oracle():
    """tools/fuzz_humanoid.py: 30 random cases -- every motion of the clip set at a random time, joints / orientation / velocities
    perturbed around the clip's pose (every third case far: folded limbs, self-collision, tendon limits), all three spline
    representations -- 8 candidates x 40 steps each on rollout_tree_kernel<Humanoid> (fp64) against the oracle: failure flags equal,
    states and residuals of the first steps within 1e-9. 90 such cases: profiles/r03_fuzz_humanoid.log (2e-11 over 40 steps)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_humanoid.py"), "30", "3"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "30 cases x 8 candidates" in out.stdout.splitlines()[-1]
