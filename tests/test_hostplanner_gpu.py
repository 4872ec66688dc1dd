"""-m gpu: the C++ planner (driven through hostplanner.py) against the Python mirror, and its sharded mode:
two processes sharing ONE GPU (gloo as the exchange transport) must reproduce the single-rank plan exactly."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_planner_equals_python_planner(cartpole):
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuSamplingPlanner, State
    H, N = 40, 512
    cpp = HostPlanner(cartpole, seed=3, num_trajectory=N)
    cpp.reset(H)
    py = GpuSamplingPlanner(seed=3)
    py.initialize(cartpole.model, cartpole); py.num_trajectory_ = N; py.allocate(); py.reset(H)
    st = State(cartpole.model)
    for k in range(5):
        q, v, t = [0.1 * k, 0.5], [0.0, -0.1], 0.04 * k
        st.set(q, v, time=t); py.set_state(st); py.optimize_policy(H)
        cpp.set_state(q, v, t); cpp.optimize_policy(H)
        assert cpp.winner == py.winner
        assert cpp.best_score == py.candidate_score(0) and cpp.improvement == py.improvement
        ct, cv = cpp.policy()
        assert np.array_equal(ct, py.policy.plan.times()) and np.array_equal(cv, py.policy.plan.values())
        a = np.zeros(1)
        py.action_from_policy(a, None, t + 0.013)
        assert np.array_equal(cpp.action(t + 0.013), a)
    assert "StaticCartpole" in cpp.kernel_name


def test_cpp_cross_entropy_equals_python_planner(particle):
    """mjpc::GpuCrossEntropyPlanner (C++) against planners.GpuCrossEntropyPlanner (itself checked against the numpy
    restatement of cross_entropy/planner.cc and the oracle backend in tests/test_cross_entropy.py)."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuCrossEntropyPlanner, State
    H, N = 30, 600
    cpp = HostPlanner(particle, seed=5, num_trajectory=N, kind="cross_entropy")
    cpp.reset(H)
    py = GpuCrossEntropyPlanner(seed=5)
    py.initialize(particle.model, particle); py.num_trajectory_ = N; py.n_elite_ = max(N // 10, 2); py.allocate(); py.reset(H)
    st = State(particle.model)
    P, nu = py.policy.num_spline_points, particle.model.nu
    for k in range(5):
        q, v, t = [0.02 * k, -0.05], [0.0, 0.1], 0.05 * k
        st.set(q, v, time=t); py.set_state(st); py.optimize_policy(H)
        cpp.set_state(q, v, t); cpp.optimize_policy(H)
        assert list(cpp.ce_elites()) == py.trajectory_order
        assert cpp.improvement == py.improvement
        ct, cv = cpp.policy()
        assert np.array_equal(ct, py.policy.plan.times()) and np.array_equal(cv, py.policy.plan.values())
        assert np.array_equal(cpp.ce_variance(P * nu), py.variance[:P * nu])
        a = np.zeros(nu)
        py.action_from_policy(a, None, t + 0.013)
        assert np.array_equal(cpp.action(t + 0.013), a)
    bt, pt = cpp.best_trajectory(), py.best_trajectory()
    assert bt["total_return"] == pt.total_return and np.array_equal(bt["states"], pt.states[:H])


def test_cpp_cross_entropy_on_the_quadruped_equals_python_planner():
    """BASELINE configs[2] through the C++ host: mjpc::QuadrupedFlat (ResetLocked ids, Transition state, frozen residual
    copy) + mjpc::GpuCrossEntropyPlanner on the wavefront-per-candidate kernel == the Python mirror, bit for bit."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuCrossEntropyPlanner, State
    from mujoco_mpc_amd.task import load_task
    t = load_task("QuadrupedFlat")
    t.parameters[t.ids["gait"]] = 2.0  # Trot: exercises the gait switch in Transition (weights and parameters change)
    t.transition(0.0)
    H, N = 20, 128
    cpp = HostPlanner(load_task("QuadrupedFlat"), seed=9, num_trajectory=N, kind="cross_entropy")
    cpp.task_set_parameter(t.ids["gait"], 2.0)
    cpp.task_transition(0.0)
    cpp.reset(H)
    py = GpuCrossEntropyPlanner(seed=9)
    py.initialize(t.model, t); py.num_trajectory_ = N; py.n_elite_ = max(N // 10, 2); py.allocate(); py.reset(H)
    st = State(t.model)
    home = t.model.keyframes["home"]["qpos"]
    mp = np.array([[0.3, 0, 0.26], [-2.5, 0, 0]]); mq = np.array([[1.0, 0, 0, 0], [1.0, 0, 0, 0]])
    for k in range(2):
        tm = 0.01 * k
        st.set(home, np.zeros(18), mocap_pos=mp, mocap_quat=mq, time=tm); py.set_state(st); py.optimize_policy(H)
        cpp.set_state(home, np.zeros(18), tm, mocap_pos=mp, mocap_quat=mq); cpp.optimize_policy(H)
        assert list(cpp.ce_elites()) == py.trajectory_order
        ct, cv = cpp.policy()
        assert np.array_equal(ct, py.policy.plan.times()) and np.array_equal(cv, py.policy.plan.values())
        assert cpp.improvement == py.improvement
    assert "rollout_wave_kernel" in cpp.kernel_name or "rollout_tree_kernel" in cpp.kernel_name


def test_cpp_cross_entropy_at_full_size_equals_python_planner():
    """BASELINE configs[2] at FULL size through the C++ host: 16384 candidates x horizon 100, n_elite = 1638, three plan iterations of
    mjpc::GpuCrossEntropyPlanner on rollout_quad_kernel == the Python mirror bit for bit (elite order, policy, improvement)."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuCrossEntropyPlanner, State
    from mujoco_mpc_amd.task import load_task
    t = load_task("QuadrupedFlat")
    t.transition(0.0)
    H, N = 100, 16384
    cpp = HostPlanner(load_task("QuadrupedFlat"), seed=3, num_trajectory=N, kind="cross_entropy")
    cpp.task_transition(0.0)
    cpp.reset(H)
    py = GpuCrossEntropyPlanner(seed=3)
    py.initialize(t.model, t); py.num_trajectory_ = N; py.n_elite_ = N // 10; py.allocate(); py.reset(H)
    assert py.n_elite_ == 1638
    st = State(t.model)
    home = t.model.keyframes["home"]["qpos"]
    mp = np.array([[0.3, 0, 0.26], [-2.5, 0, 0]]); mq = np.array([[1.0, 0, 0, 0], [1.0, 0, 0, 0]])
    for k in range(3):
        tm = 0.01 * k
        st.set(home, np.zeros(18), mocap_pos=mp, mocap_quat=mq, time=tm); py.set_state(st); py.optimize_policy(H)
        cpp.set_state(home, np.zeros(18), tm, mocap_pos=mp, mocap_quat=mq); cpp.optimize_policy(H)
        elites = list(cpp.ce_elites())
        assert len(elites) == 1638 and elites == py.trajectory_order
        ct, cv = cpp.policy()
        assert np.array_equal(ct, py.policy.plan.times()) and np.array_equal(cv, py.policy.plan.values())
        assert cpp.improvement == py.improvement
    assert cpp.kernel_name.startswith("rollout_quad_kernel")


def test_cpp_predictive_sampling_replans_on_the_quadruped():
    """several plan iterations through mjpcx_best on a wave-kernel context (regression: the model allocation must
    outlive the first policy update) -- the return of the winner keeps improving from a standing start"""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.task import load_task
    t = load_task("QuadrupedFlat")
    p = HostPlanner(t, seed=1, num_trajectory=64, kind="sampling")
    p.task_transition(0.0)
    H = t.planning_steps()
    p.reset(H)
    home = t.model.keyframes["home"]["qpos"]
    mp = np.array([[0.3, 0, 0.26], [-2.5, 0, 0]]); mq = np.array([[1.0, 0, 0, 0], [1.0, 0, 0, 0]])
    scores = []
    for k in range(4):
        p.set_state(home, np.zeros(18), 0.0, mocap_pos=mp, mocap_quat=mq)
        p.optimize_policy(H)
        scores.append(p.best_score)
    assert all(np.isfinite(scores)) and scores[-1] < scores[0]


@pytest.mark.parametrize("limits,reg", [(1, 0), (0, 2)])
def test_cpp_ilqg_equals_python_planner(particle, limits, reg):
    """mjpc::GpuILQGPlanner (C++) against planners.GpuILQGPlanner (itself checked against the oracle backend in
    tests/test_ilqg_planner.py): same device calls in the same order => identical policies."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuILQGPlanner, State
    H = 25
    cpp = HostPlanner(particle, kind="ilqg")
    cpp.reset(H)
    cpp.ilqg_set(regularization_type=reg, action_limits=limits)
    py = GpuILQGPlanner()
    py.initialize(particle.model, particle); py.allocate(); py.reset(H)
    py.settings.regularization_type, py.settings.action_limits = reg, limits
    st = State(particle.model)
    for k in range(4):
        q, v, t = [0.03 * k, -0.04], [0.1, 0.0], 0.05 * k
        st.set(q, v, time=t); py.set_state(st); py.optimize_policy(H)
        cpp.set_state(q, v, t); cpp.optimize_policy(H)
        info = cpp.ilqg_info()
        assert info["winner"] == py.winner and info["regularization"] == py.regularization
        assert info["dV0"] == py.dV[0] and info["dV1"] == py.dV[1] and info["improvement"] == py.improvement
        ct, cx, cu, cK = cpp.ilqg_policy(H)
        tr = py.policy.trajectory
        assert np.array_equal(ct, tr.times[:H]) and np.array_equal(cx, tr.states[:H]) and np.array_equal(cu, tr.actions[:H])
        assert np.array_equal(cK, py.policy.feedback_gain[:H])
        a = np.zeros(particle.model.nu)
        x = np.array([0.03 * k + 0.01, -0.03, 0.1, 0.02])
        py.action_from_policy(a, x, t + 0.013)
        np.testing.assert_allclose(cpp.action(t + 0.013, state=x), a, rtol=1e-13, atol=1e-15)  # K dx: numpy dot vs C loop


WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from mujoco_mpc_amd.distributed import RankGroup
from mujoco_mpc_amd.hostplanner import HostPlanner
from mujoco_mpc_amd.task import load_task
world = int(os.environ.get("WORLD_SIZE", "1"))
group = None
if world > 1:
    dist.init_process_group(backend="gloo")
    group = RankGroup(dist, torch.device("cpu"))
kind = os.environ.get("MJPC_TEST_KIND", "sampling")
if kind == "quadruped":   # the A1 at a batch whose two-rank shares sit AT the quad kernel's threshold: both world sizes must run the same kernel
    task = load_task("QuadrupedFlat")
    p = HostPlanner(task, device=0, seed=7, num_trajectory=4096, group=group, kind="sampling")
    H = 12
    p.task_transition(0.0)
    p.reset(H)
    home = task.model.keyframes["home"]["qpos"]
    log = []
    for k in range(3):
        p.set_state(home, np.zeros(18), 0.0, mocap_pos=np.array([[0.3, 0, 0.26], [-2.5, 0, 0]]), mocap_quat=np.array([[1, 0, 0, 0], [1, 0, 0, 0.]]))
        p.optimize_policy(H)
        t, v = p.policy()
        log.append(dict(winner=p.winner, score=p.best_score, improvement=p.improvement, plan=v.tolist(), kernel=p.kernel_name))
    if group is None or group.rank == 0:
        print("RESULT " + json.dumps(log))
    if group is not None:
        dist.barrier(); dist.destroy_process_group()
    sys.exit(0)
task = load_task("Cartpole")
p = HostPlanner(task, device=0, seed=7, num_trajectory=1000, group=group, kind=kind)
H = 32
if kind == "robust":
    p.robust_config(ncandidates=9, nrepetitions=3, xfrc_std=0.2, xfrc_rate=0.1)
p.reset(H)
log = []
for k in range(4):
    p.set_state([0.05 * k, 0.3], [0.0, 0.1], 0.04 * k)
    p.optimize_policy(H)
    t, v = p.policy()
    rec = dict(winner=p.winner, score=p.best_score, improvement=p.improvement, plan=v.tolist())
    if kind == "robust":
        best, scores = p.robust_result(9)
        rec.update(best_candidate=best, perturbed=scores.tolist())
    if kind == "cross_entropy":
        rec["elites"] = p.ce_elites().tolist()
        rec["variance"] = p.ce_variance(v.size).tolist()
    log.append(rec)
if group is None or group.rank == 0:
    print("RESULT " + json.dumps(log))
if group is not None:
    dist.barrier(); dist.destroy_process_group()
'''


def run(world, kind="sampling"):
    script = WORKER % dict(root=ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MJPC_TEST_KIND=kind)
    cmd = [sys.executable, "-c", script] if world == 1 else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
        "127.0.0.1", "--master-port", "29544", "--no-python", sys.executable, "-c", script]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_two_ranks_on_one_gpu_equal_one_rank():
    one, two = run(1), run(2)
    assert one == two


def test_two_ranks_on_one_gpu_equal_one_rank_on_the_quadruped():
    """4096 candidates of the A1: one rank rolls them out on rollout_quad_kernel, two ranks 2048 each -- on the same kernel (the hand-over
    threshold is a rank's share of BASELINE configs[2]'s 16384 over 8 GPUs), so winner, score and plan are equal bit for bit, not to a
    tolerance (results independent of the rank count)."""
    one, two = run(1, "quadruped"), run(2, "quadruped")
    assert one == two
    assert all("rollout_quad_kernel" in r["kernel"] for r in one + two)


def test_cross_entropy_two_ranks_on_one_gpu_equal_one_rank():
    """elite set identical; the moments are all-reduced partial sums, so mean/variance agree to rounding"""
    one, two = run(1, "cross_entropy"), run(2, "cross_entropy")
    for a, b in zip(one, two):
        assert a["elites"] == b["elites"]
        np.testing.assert_allclose(a["plan"], b["plan"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(a["variance"], b["variance"], rtol=1e-10, atol=1e-16)
        np.testing.assert_allclose(a["improvement"], b["improvement"], rtol=1e-10, atol=1e-14)


def test_robust_planner_two_ranks_on_one_gpu_equal_one_rank():
    """GpuRobustPlanner sharded: the delegate's 1000 candidates 500 + 500, the 9 best of both ranks merged (with their splines), the
    27 perturbed rollouts 14 + 13 with the force noise keyed on the global rollout index -- same ranked candidates, same perturbed
    means, same policy as one rank, bit for bit (the exchanges move numbers, they do not re-associate sums)."""
    one, two = run(1, "robust"), run(2, "robust")
    assert one == two
    assert all(0 <= r["best_candidate"] < 9 and len(set(r["perturbed"])) == 9 for r in one)


def test_cpp_robust_planner_on_the_quadruped():
    """GpuRobustPlanner (robust_planner.cc:90-170): the delegate's best candidates are re-rolled under xfrc_applied noise in
    one launch; the winner has the lowest mean perturbed return, its spline becomes the policy, and that mean is what the
    oracle's NoisyRollout gives for the same spline and noise stream."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.task import load_task
    from oracle import pyoracle
    t = load_task("QuadrupedFlat")
    p = HostPlanner(t, seed=2, num_trajectory=64, kind="robust")
    K, R = 4, 3
    p.robust_config(ncandidates=K, nrepetitions=R, xfrc_std=0.3, xfrc_rate=0.1)
    p.task_transition(0.0)
    t.transition(0.0)
    H = t.planning_steps()
    p.reset(H)
    home = t.model.keyframes["home"]["qpos"]
    mp = np.array([[0.3, 0, 0.26], [-2.5, 0, 0]]); mq = np.array([[1.0, 0, 0, 0], [1.0, 0, 0, 0]])
    p.set_state(home, np.zeros(18), 0.0, mocap_pos=mp, mocap_quat=mq)
    p.optimize_policy(H)
    best, scores = p.robust_result()
    assert 0 <= best < K and scores[best] == scores[:K].min() and np.all(scores[:K] > 0)
    times, values = p.policy()
    mocap = np.hstack([mp, mq]).reshape(-1)
    state = np.concatenate([home, np.zeros(18)])
    from mujoco_mpc_amd import capi
    ref = pyoracle.rollout_batch(t.packed_model(), t.packed(), state, 0.0, mocap, R, H, len(times), capi.SPLINE_CUBIC, times,  # PS default
                                 np.tile(values, (R, 1, 1)), num_threads=2, full=False, xfrc_std=0.3, xfrc_rate=0.1, seed=2,
                                 candidate_offset=best * R)
    assert not ref["failure"].any()
    # the reference's running mean over the valid perturbed rollouts = their plain mean
    assert abs(ref["total_return"].mean() - scores[best]) < 1e-7 * (1 + abs(scores[best]))
    # a second plan draws fresh noise (the candidate offset advances with the iteration) and still runs
    p.optimize_policy(H)
    assert 0 <= p.robust_result()[0] < K


def test_cpp_sample_gradient_planner(cartpole):
    """GpuSampleGradientPlanner (sample_gradient/planner.cc): first iteration checked against a by-hand evaluation of the
    fitness-shaped gradient from the SAME noise stream (host Philox = device = oracle) and the returned rollout costs;
    then it keeps improving the plan and uses its gradient candidates."""
    from mujoco_mpc_amd.hostplanner import HostPlanner, host_gaussian_pair
    N, G, H = 24, 4, 40
    p = HostPlanner(cartpole, seed=7, num_trajectory=N, kind="sample_gradient")
    p.sample_gradient_config(num_gradient=G, gradient_filter=0.8)
    p.reset(H)
    P, nu = p.num_spline_points, 1
    q, v = [0.2, 0.6], [0.0, 0.1]
    p.set_state(q, v, 0.0)
    p.optimize_policy(H)
    wt, g, ret = p.sample_gradient_result(P * nu, N)
    num_noisy = N - G
    assert np.all(ret[:N] > 0) and wt in (0, 1, 2)
    # noise of candidate i, iteration 0: pairs (k >> 1) of the candidate's stream
    noise = np.zeros((num_noisy, P * nu))
    for i in range(1, num_noisy):
        for k in range(0, P * nu, 2):
            z = host_gaussian_pair(7, i, k >> 1, 0)
            noise[i, k] = z[0]
            if k + 1 < P * nu:
                noise[i, k + 1] = z[1]
    order = np.argsort(ret[:num_noisy], kind="stable")
    f0 = np.log(0.5 * num_noisy + 1.0)
    raw = np.maximum(0.0, f0 - np.log(order + 1.0))      # the reference indexes the shaping by trajectory_order[i]
    w = raw / raw.sum() - 1.0 / num_noisy
    expect = sum(noise[order[i]] * (w[i] / num_noisy) for i in range(num_noisy))
    assert np.allclose(g, expect, rtol=1e-12, atol=1e-15)
    # the gradient candidates of iteration 0 (zeros + steps along -gradient) are rolled out at iteration 1
    scores = [ret[p.winner]]
    types = [wt]
    for k in range(1, 12):
        p.set_state(q, v, 0.0)
        p.optimize_policy(H)
        wt, g, ret = p.sample_gradient_result(P * nu, N)
        scores.append(ret[p.winner]); types.append(wt)
        assert p.best_score == ret[p.winner]
    assert scores[-1] < scores[0] and all(b <= a + 1e-12 for a, b in zip(scores, scores[1:]))   # the nominal is always a candidate
    assert 2 in types or 1 in types
    tr = p.best_trajectory()
    assert abs(tr["total_return"] - scores[-1]) < 1e-12


def test_cpp_robust_planner_on_the_cartpole():
    """the same planner on a lane-per-candidate model (NoisyRollout in rollout_lane_kernel<..., NOISY = true>)"""
    from mujoco_mpc_amd import capi
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.task import load_task
    from oracle import pyoracle
    t = load_task("Cartpole")
    p = HostPlanner(t, seed=4, num_trajectory=128, kind="robust")
    K, R = 5, 4
    p.robust_config(ncandidates=K, nrepetitions=R, xfrc_std=0.5, xfrc_rate=0.05)
    H = t.planning_steps()
    p.reset(H)
    p.set_state([0.4, 2.6], [0.1, -0.3], 0.0)
    p.optimize_policy(H)
    best, scores = p.robust_result()
    assert 0 <= best < K and scores[best] == scores[:K].min() and np.all(scores[:K] > 0)
    times, values = p.policy()
    ref = pyoracle.rollout_batch(t.packed_model(), t.packed(), [0.4, 2.6, 0.1, -0.3], 0.0, None, R, H, len(times), capi.SPLINE_CUBIC,
                                 times, np.tile(values, (R, 1, 1)), num_threads=2, xfrc_std=0.5, xfrc_rate=0.05, seed=4,
                                 candidate_offset=best * R)
    assert abs(ref["total_return"].mean() - scores[best]) <= 1e-9 * (1 + abs(scores[best]))
    p.close()


def test_native_rccl_exchange_on_one_rank():
    """mjpcx_comm_* (include/mjpcx.h): a one-rank RCCL communicator on this GPU -- every collective the sharded planners use
    goes through librccl (all-gather, broadcast, all-reduce) and must leave a single rank's records unchanged. (More than one
    rank needs more than one GPU: RCCL refuses two ranks on one device. The exchange LOGIC for several ranks is covered by the
    gloo test through the planners' transport callbacks, tests/test_distributed_gloo.py.)"""
    import ctypes as C
    from mujoco_mpc_amd import capi
    from mujoco_mpc_amd.hostplanner import comm_unique_id
    from mujoco_mpc_amd.task import load_task
    L = capi.lib()
    L.mjpcx_last_error.restype = C.c_char_p
    t = load_task("Particle")
    ctx = capi.Context(t.packed_model(), t.packed(), 0, 64)
    uid = C.create_string_buffer(comm_unique_id(), 128)
    assert L.mjpcx_comm_init(ctx.handle, uid, 0, 1) == 0, L.mjpcx_last_error(ctx.handle)
    rank, world = C.c_int(-1), C.c_int(-1)
    assert L.mjpcx_comm_info(ctx.handle, C.byref(rank), C.byref(world)) == 0 and (rank.value, world.value) == (0, 1)
    idx, best, nominal = C.c_int32(17), C.c_double(0.25), C.c_double(0.5)
    vals = (C.c_double * 6)(1, 2, 3, 4, 5, 6)
    assert L.mjpcx_exchange_best(ctx.handle, C.byref(idx), C.byref(best), C.byref(nominal), vals, 6) == 0, L.mjpcx_last_error(ctx.handle)
    assert (idx.value, best.value, nominal.value, list(vals)) == (17, 0.25, 0.5, [1, 2, 3, 4, 5, 6])
    k = 4
    index = (C.c_int64 * k)(9, 3, 12, -1)
    ret = (C.c_double * k)(0.5, 0.5, 0.1, 1e300)
    assert L.mjpcx_merge_topk(ctx.handle, k, index, ret) == 0, L.mjpcx_last_error(ctx.handle)
    assert list(index) == [12, 3, 9, -1] and list(ret)[:3] == [0.1, 0.5, 0.5]   # sorted by return, ties by global index
    v = (C.c_double * 3)(1.5, -2.0, 4.0)
    assert L.mjpcx_elite_allreduce(ctx.handle, v, 3) == 0 and list(v) == [1.5, -2.0, 4.0]
    assert L.mjpcx_comm_barrier(ctx.handle) == 0
    assert L.mjpcx_comm_destroy(ctx.handle) == 0
    ctx.close()


def test_cpp_planner_with_a_native_one_rank_communicator(particle):
    """the C++ planner's sharded code path with the library's own exchange (no transport callback): world = 1 == unsharded"""
    from mujoco_mpc_amd.hostplanner import HostPlanner, comm_unique_id
    ref = HostPlanner(particle, seed=3, num_trajectory=64)
    nat = HostPlanner(particle, seed=3, num_trajectory=64, native_comm=(comm_unique_id(), 0, 1))
    for p in (ref, nat):
        p.reset(11)
        p.set_state([0.02, -0.05], [0.0, 0.1])
    for _ in range(3):
        ref.optimize_policy(11)
        nat.optimize_policy(11)
        assert ref.winner == nat.winner and ref.best_score == nat.best_score
    assert np.array_equal(ref.policy()[1], nat.policy()[1])
    nat.comm_barrier()


def test_agent_integrator_rk4_reaches_the_device_through_the_cpp_planner():
    """`<numeric name="agent_integrator" data="1"/>` (mjINT_RK4) in a task: Agent::PlanIteration plans on a model copy with that
    integrator (agent.cc:288-291). The C++ planner passes it through FlatModel to the device; plans equal the Python mirror's, which
    reads the same numeric, and differ from the Euler plans of the same seed."""
    import copy
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuSamplingPlanner, State
    from mujoco_mpc_amd.task import load_task
    euler = load_task("Cartpole")
    rk4 = load_task("Cartpole")
    rk4.model = copy.deepcopy(rk4.model)
    rk4.model.numeric["agent_integrator"] = [1.0]
    assert rk4.packed_model().struct.integrator == 1 and euler.packed_model().struct.integrator == 0
    H, N = 40, 256
    plans = {}
    for name, task in (("euler", euler), ("rk4", rk4)):
        cpp = HostPlanner(task, seed=3, num_trajectory=N)
        cpp.reset(H)
        py = GpuSamplingPlanner(seed=3)
        py.initialize(task.model, task); py.num_trajectory_ = N; py.allocate(); py.reset(H)
        st = State(task.model)
        for k in range(3):
            q, v, t = [0.1 * k, 0.5], [0.0, -0.1], 0.04 * k
            st.set(q, v, time=t); py.set_state(st); py.optimize_policy(H)
            cpp.set_state(q, v, t); cpp.optimize_policy(H)
            assert cpp.winner == py.winner and cpp.best_score == py.candidate_score(0)
        plans[name] = (cpp.best_score, cpp.policy()[1].copy())
        cpp.close()
    assert plans["euler"][0] != plans["rk4"][0]


def test_a_missing_rank_does_not_hang_the_communicator():
    """mjpcx_comm_init with a world of 2 and nobody else joining: ncclCommInitRank would block for good; the library gives up after
    MJPCX_COMM_TIMEOUT_S and reports MJPCX_EDEVICE, so a planner can fall back to its own transport. (In a child process: the abandoned
    RCCL bootstrap thread stays blocked until the process ends.)"""
    import subprocess
    import sys
    code = r'''
import ctypes as C, os, sys, time
os.environ["MJPCX_COMM_TIMEOUT_S"] = "4"
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.hostplanner import comm_unique_id
from mujoco_mpc_amd.task import load_task
L = capi.lib()
t = load_task("Particle")
ctx = capi.Context(t.packed_model(), t.packed(), 0, 64)
uid = C.create_string_buffer(comm_unique_id(), 128)
t0 = time.time()
rc = L.mjpcx_comm_init(ctx.handle, uid, 0, 2)
dt = time.time() - t0
rank, world = C.c_int(-1), C.c_int(-1)
L.mjpcx_comm_info(ctx.handle, C.byref(rank), C.byref(world))
print("RESULT", rc, round(dt, 1), world.value, flush=True)
os._exit(0)
'''
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
    assert line, out.stdout + out.stderr
    rc, dt, world = line[0].split()[1:]
    assert int(rc) == -3 and 3.0 <= float(dt) < 30.0 and int(world) == 1   # MJPCX_EDEVICE, after the deadline, no communicator


def test_a_sharded_planner_without_a_transport_says_so(particle):
    """world > 1 with neither exchange callbacks nor a communicator of that size: the library's exchange would be the identity and every
    rank would silently keep its own best -- the planners refuse instead (MJPCX_ESTATE)"""
    import ctypes as C
    from mujoco_mpc_amd.hostplanner import EXCHANGE_FN, MERGE_TOPK_FN, SUM_FN, HostPlanner, lib
    for kind in ("sampling", "cross_entropy"):
        p = HostPlanner(particle, seed=3, num_trajectory=64, kind=kind)
        if kind == "sampling":
            assert lib().mjpc_planner_set_sharding(p.h, 0, 2, C.cast(None, EXCHANGE_FN), None) == 0
        else:
            assert lib().mjpc_planner_set_sharding_ce(p.h, 0, 2, C.cast(None, MERGE_TOPK_FN), C.cast(None, SUM_FN), None) == 0
        p.reset(10)
        p.set_state(np.zeros(2), np.zeros(2), 0.0, mocap_pos=np.array([[0.2, 0.1, 0.01]]), mocap_quat=np.array([[1.0, 0, 0, 0]]))
        with pytest.raises(RuntimeError, match="communicator of its world size"):
            p.optimize_policy(10)
        p.close()
