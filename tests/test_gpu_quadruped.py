"""-m gpu: the wavefront-per-candidate kernel (free joint, friction loss, elliptic-cone contacts, Newton solver, the
QuadrupedFlat residual) against the CPU oracle on the Unitree A1 of BASELINE configs[2].

Tolerance: contact-rich dynamics amplify rounding differences between the two implementations (different summation
orders in the subtree sums, FMA contraction, device libm), so the bound loosens with the horizon:
|gpu - oracle| <= tol (1 + |oracle|) with tol = 1e-9 for the first steps and 1e-6 over 40 steps (observed ~1e-10)."""
import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

pytestmark = pytest.mark.gpu
MOCAP = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0])


@pytest.fixture(scope="module")
def quad():
    t = load_task("QuadrupedFlat")
    t.transition(0.0)
    return t


def close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= tol * (1 + np.abs(b)))


def run(task, state, N, H, P, interp, seed, tol, time=0.0, integrator=None):
    pm, pt = task.packed_model(), task.packed()
    if integrator is not None:
        pm.struct.integrator = integrator
    rng = np.random.default_rng(seed)
    dt = task.model.get_number("agent_timestep", task.model.timestep)
    times = time + np.arange(P) * max((H - 1) * dt / max(P - 1, 1), 1e-3)
    nodes = np.clip(rng.normal(0, 0.4, (N, P, task.model.nu)), -1, 1)
    ctx = capi.Context(pm, pt, 0, 64)
    assert "rollout_wave_kernel" in ctx.kernel_name or "rollout_tree_kernel" in ctx.kernel_name
    ctx.set_state(state, time, MOCAP)
    ctx.rollout_splines(H, interp, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, time, MOCAP, N, H, P, interp, times, nodes, num_threads=8)
    assert np.array_equal(fail, ref["failure"]) and not fail.any()
    worst = 0.0
    for c in range(N):
        tr = ctx.fetch_trajectory(c)
        for name in ("states", "actions", "times", "residual", "costs", "trace"):
            g, o = getattr(tr, name), ref[name][c]
            worst = max(worst, float(np.max(np.abs(g - o) / (1 + np.abs(o)))))
            assert close(g, o, tol), (name, c, float(np.max(np.abs(g - o))))
    assert close(ret, ref["total_return"], tol)
    ctx.close()
    return worst


def test_standing_start_short(quad):
    home = quad.model.keyframes["home"]["qpos"]
    state = np.concatenate([home, np.zeros(18)])
    run(quad, state, N=6, H=6, P=3, interp=0, seed=1, tol=1e-9)


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_forty_steps(quad, interp):
    home = quad.model.keyframes["home"]["qpos"]
    state = np.concatenate([home, np.zeros(18)])
    run(quad, state, N=8, H=40, P=4, interp=interp, seed=2 + interp, tol=1e-6)


def test_falling_and_tumbling(quad):
    """dropped from 0.5 m with spin: body geoms hit the floor, many contacts, row cap exercised"""
    q = quad.model.keyframes["home"]["qpos"].copy()
    q[2] = 0.5
    q[3:7] = [0.9, 0.3, 0.2, 0.1]
    q[3:7] /= np.linalg.norm(q[3:7])
    v = np.zeros(18)
    v[3:6] = [2.0, -1.0, 0.5]
    run(quad, np.concatenate([q, v]), N=4, H=60, P=3, interp=0, seed=5, tol=1e-5)


def test_rollout_feedback_on_the_three_kernels_agree(quad, monkeypatch):
    """The A1's feedback rollouts run on rollout_feedback_quad_kernel (four lanes per candidate, one candidate per wavefront: the shortest
    step of the three); MJPCX_NO_QUAD_FEEDBACK=1 keeps them on rollout_feedback_tree_kernel<A1> (model image and plan blob in LDS),
    MJPCX_NO_LDS_MODEL=1 on the generic kernel (model behind global pointers). Three kernels, the same numbers; and a candidate the quad
    form hands on (MJPCX_QUAD_CON_CAP=1: every lane with two contacts) comes back from the wavefront-per-candidate kernel."""
    H = 24
    pm, pt, nom, state = nominal_quad(quad, H, 21)
    rng = np.random.default_rng(22)
    gains = 0.05 * rng.normal(size=(H, 12, 36))
    improvement = 0.05 * rng.normal(size=(H, 12))
    alpha = np.concatenate([np.exp(np.linspace(0, np.log(1e-3), 9)), [0.0]])
    start = state.copy()
    start[19:] = 0.05 * rng.normal(size=18)
    got = []
    for env in ({}, {"MJPCX_NO_QUAD_FEEDBACK": "1"}, {"MJPCX_NO_LDS_MODEL": "1"}, {"MJPCX_QUAD_CON_CAP": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = capi.Context(pm, pt, 0, 64)
        assert ctx.kernel_name.startswith("rollout_wave_kernel" if "MJPCX_NO_LDS_MODEL" in env else "rollout_quad_kernel")   # (the latter: a registered A1)
        ctx.set_state(start, 0.0, MOCAP)
        for mode, rep in ((0, 0), (1, 2)):
            ctx.rollout_feedback(H, mode, rep, 1, nom["times"], nom["states"], nom["actions"], gains, improvement, alpha)
            ret, fail = ctx.returns()
            tr = ctx.fetch_trajectory(3)
            got.append((ret.copy(), fail.copy(), tr.states.copy(), tr.actions.copy(), tr.residual.copy(), tr.costs.copy()))
            if "MJPCX_QUAD_CON_CAP" in env:
                assert ctx.quad_stats()["handed_on"] == len(alpha)   # (every candidate has a lane with two contacts at some step)
            elif not env:   # (random gains of 0.05 push a joint of the wilder candidates out of the range the pair proofs cover: handed on, legitimately)
                st = ctx.quad_stats()
                assert st["handed_on"] < len(alpha) // 2 and st["handed_on"] == st["out_of_proof_range"], st
        ctx.close()
        for k in env:
            monkeypatch.delenv(k)
    for variant in (got[2:4], got[4:6], got[6:8]):
        for a, b in zip(got[:2], variant):
            assert not a[1].any() and not b[1].any()
            for x, y in zip(a, b):
                assert close(x, y, 1e-11), float(np.abs(x - y).max())


@pytest.mark.parametrize("tree", [True, False])
def test_rk4_integrator_on_the_wave_kernels(quad, tree, monkeypatch):
    """mjINT_RK4 on the A1: four forward passes (contacts, friction loss, the Newton solver warm-started from the previous step in
    every stage) per mj_step, on the Jacobian-free kernel and on the row-table kernel, against oracle/physics.c o_rk4"""
    if not tree:
        monkeypatch.setenv("MJPCX_NO_TREE", "1")
    home = quad.model.keyframes["home"]["qpos"]
    v = np.zeros(18); v[0:3] = [0.3, 0.0, -0.2]; v[3:6] = [0.5, -0.4, 0.3]
    # (seed 12: at most 51 constraint rows per step -- the row-table kernel stages 64; seed 11 reaches 66 since the hips' cylinders collide)
    run(quad, np.concatenate([home, v]), N=5, H=30, P=3, interp=2, seed=12, tol=1e-6, integrator=1)


def test_trot_gait_residual(quad):
    t = load_task("QuadrupedFlat")
    t.parameters[t.ids["gait"]] = 2.0   # Trot
    t.transition(0.0)
    home = t.model.keyframes["home"]["qpos"]
    run(t, np.concatenate([home, np.zeros(18)]), N=4, H=20, P=3, interp=0, seed=7, tol=1e-7, time=0.0)


@pytest.mark.parametrize("mode,biped_type,gait,flip_dir", [(1, 0, 0, 0), (1, 1, 0, 0), (2, 0, 1, 0), (3, 0, 3, 0), (4, 0, 0, 0), (4, 0, 0, 1),
                                                           (0, 0, 4, 0)])
def test_every_residual_mode_on_the_device(mode, biped_type, gait, flip_dir):
    """QuadrupedFlat::ResidualFn::Residual (quadruped.cc:33-226) has five modes -- Quadruped, Biped (feet or handstand), Walk,
    Scramble, Flip -- and five gaits; the frozen ResidualFn state selects the branch. Each branch on the device against the oracle
    (residual entries, costs, returns), with the task state a Transition into that mode leaves (mode start time, a non-trivial
    saved orientation / position / heading for the Flip and Walk branches)."""
    t = load_task("QuadrupedFlat")
    t.parameters[t.ids["gait"]] = float(gait)
    t.parameters[t.ids["biped_type"]] = float(biped_type)
    t.parameters[t.ids["flip_dir"]] = float(flip_dir)
    t.transition(0.0)
    t.mode = t.current_mode = mode
    t.mode_start_time = 0.02
    t.position, t.heading_vec, t.speed, t.angvel, t.ground = [0.05, -0.02, 0.27], [0.9, 0.1], 0.3, 0.2, 0.0
    t.orientation = [0.995, 0.02, 0.05, 0.08]
    t._freeze()
    home = t.model.keyframes["home"]["qpos"].copy()
    v = np.zeros(18)
    v[0:3] = [0.2, -0.1, 0.0]
    v[3:6] = [0.1, 0.3, -0.2]
    worst = run(t, np.concatenate([home, v]), N=4, H=24, P=3, interp=1, seed=40 + mode, tol=1e-7, time=0.05)
    assert worst < 1e-7


def test_cross_entropy_planner_on_the_quadruped(quad):
    """BASELINE configs[2] in miniature: Cross-Entropy on the A1 (zero-order splines, 3 nodes) closes the loop with the
    oracle as the plant for a few steps; the planned return improves on the zero policy and the robot stays up."""
    from mujoco_mpc_amd.planners import GpuCrossEntropyPlanner, State
    t = load_task("QuadrupedFlat")
    t.transition(0.0)
    H = t.planning_steps()
    p = GpuCrossEntropyPlanner(seed=1)
    p.initialize(t.model, t)
    p.num_trajectory_, p.n_elite_ = 256, 26
    p.allocate()
    p.reset(H)
    st = State(t.model)
    ph = pyoracle.Physics(t.packed_model(planning=False))
    q, v = t.model.keyframes["home"]["qpos"].copy(), np.zeros(18)
    st.set(q[:19], v, mocap_pos=MOCAP.reshape(2, 7)[:, :3], mocap_quat=MOCAP.reshape(2, 7)[:, 3:], time=0.0)
    p.set_state(st)
    best = []
    for _ in range(4):
        p.optimize_policy(H)
        best.append(p.ctx.returns()[0][p.trajectory_order[0]])
    assert best[-1] < best[0]                       # the elite distribution tightens around better plans
    p.nominal_trajectory(H)
    nominal = p.best_trajectory()
    assert not nominal.failure and nominal.total_return < 1.5 * best[-1] + 0.05
    assert np.all(np.isfinite(p.policy.plan.values())) and np.all(np.isfinite(p.variance[:36]))


# ---------------------------------------------------------------------------------- iLQG pieces on the A1 (configs[4])
def nominal_quad(task, H, seed):
    """a nominal trajectory from the oracle (zero-order spline rollout from the home keyframe)"""
    pm, pt = task.packed_model(), task.packed()
    rng = np.random.default_rng(seed)
    home = task.model.keyframes["home"]["qpos"]
    state = np.concatenate([home, np.zeros(18)])
    times = np.arange(4) * (H - 1) * 0.01 / 3
    nodes = np.clip(rng.normal(0, 0.15, (1, 4, 12)), -1, 1)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, 1, H, 4, 1, times, nodes, num_threads=1)
    return pm, pt, {k: v[0] for k, v in ref.items() if k not in ("total_return", "failure")}, state


@pytest.mark.parametrize("centered", [0, 1])
def test_transition_fd_on_the_quadruped(quad, centered):
    """mjd_transitionFD with tangent-space perturbations (free joint) and contacts; the difference quotient amplifies the
    ~1e-13 agreement of the two step functions by 1/eps = 1e6"""
    H = 8
    pm, pt, nom, state = nominal_quad(quad, H, 11)
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.0, MOCAP)
    A, B, C, D = ctx.transition_fd(nom["times"], nom["states"], nom["actions"], 1e-6, centered)
    Ao, Bo, Co, Do = pyoracle.transition_fd(pm, pt, nom["states"], nom["times"], nom["actions"], 1e-6, centered, mocap=MOCAP)
    assert A.shape == (H, 36, 36) and B.shape == (H, 36, 12) and C.shape == (H, 42, 36) and D.shape == (H, 42, 12)
    for g, o in ((A, Ao), (B, Bo), (C, Co), (D, Do)):
        assert close(g, o, 5e-6), float(np.abs(g - o).max())
    # structure: d(next position)/d(velocity) = h on the free joint's translation; Effort residual = 0.02 * 40 * ctrl
    assert np.allclose(A[0, 0:3, 18:21], 0.01 * np.eye(3), atol=1e-3)
    assert np.allclose(D[0, 13:25, :], 0.8 * np.eye(12), atol=1e-5)
    ctx.close()


@pytest.mark.parametrize("mode,representation,use_state", [(0, 0, 1), (1, 1, 1), (1, 0, 0), (1, 2, 1)])
def test_rollout_feedback_on_the_quadruped(quad, mode, representation, use_state):
    """RolloutDiscrete / iLQGPolicy::Action with StateDiff on the free joint's quaternion"""
    H = 20
    pm, pt, nom, state = nominal_quad(quad, H, 12)
    rng = np.random.default_rng(13)
    gains = 0.05 * rng.normal(size=(H, 12, 36))
    improvement = 0.05 * rng.normal(size=(H, 12))
    alpha = np.concatenate([np.exp(np.linspace(0, np.log(1e-3), 9)), [0.0]])
    start = state.copy()
    start[0:3] += [0.01, -0.005, 0.004]
    q = start[3:7] + [0.0, 0.02, -0.01, 0.015]
    start[3:7] = q / np.linalg.norm(q)                                             # off-nominal orientation: quaternion StateDiff matters
    start[19:] = 0.05 * rng.normal(size=18)
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(start, 0.0, MOCAP)
    ctx.rollout_feedback(H, mode, representation, use_state, nom["times"], nom["states"], nom["actions"], gains, improvement, alpha)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_feedback(pm, pt, start, 0.0, MOCAP, H, mode, representation, use_state, nom["times"], nom["states"],
                                    nom["actions"], gains, improvement, alpha)
    assert np.array_equal(fail, ref["failure"]) and not fail.any()
    assert close(ret, ref["total_return"], 1e-7)
    for c in (0, 4, 9):
        tr = ctx.fetch_trajectory(c)
        for name in ("states", "actions", "times", "residual", "costs"):
            assert close(getattr(tr, name), ref[name][c], 1e-7), (name, c, float(np.abs(getattr(tr, name) - ref[name][c]).max()))
    if use_state:
        assert np.ptp(ret) > 1e-7
    ctx.close()


@pytest.mark.parametrize("tree", [True, False])
def test_ilqg_kernels_with_the_rk4_integrator(quad, tree, monkeypatch):
    """mjINT_RK4 in the iLQG entry points of the wavefront family: the finite-difference sweep (every perturbed column is one RK4
    step = four forward passes) and the feedback rollouts, on both forward-pass forms, against the oracle's RK4"""
    if not tree:
        monkeypatch.setenv("MJPCX_NO_TREE", "1")
    H = 8
    pm, pt = quad.packed_model(), quad.packed()
    pm.struct.integrator = 1
    rng = np.random.default_rng(21)
    home = quad.model.keyframes["home"]["qpos"]
    state = np.concatenate([home, np.zeros(18)])
    times = np.arange(4) * (H - 1) * 0.01 / 3
    nodes = np.clip(rng.normal(0, 0.15, (1, 4, 12)), -1, 1)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, 1, H, 4, 1, times, nodes, num_threads=1)
    nom = {k: v[0] for k, v in ref.items() if k not in ("total_return", "failure")}
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.0, MOCAP)
    A, B, C, D = ctx.transition_fd(nom["times"], nom["states"], nom["actions"], 1e-6, 0)
    Ao, Bo, Co, Do = pyoracle.transition_fd(pm, pt, nom["states"], nom["times"], nom["actions"], 1e-6, 0, mocap=MOCAP)
    for g, o in ((A, Ao), (B, Bo), (C, Co), (D, Do)):
        assert close(g, o, 2e-5), float(np.abs(g - o).max())
    gains = 0.05 * rng.normal(size=(H, 12, 36))
    improvement = 0.05 * rng.normal(size=(H, 12))
    alpha = np.array([1.0, 0.3, 0.0])
    start = state.copy()
    start[19:] = 0.05 * rng.normal(size=18)
    ctx.set_state(start, 0.0, MOCAP)
    ctx.rollout_feedback(H, 1, 1, 1, nom["times"], nom["states"], nom["actions"], gains, improvement, alpha)
    ret, fail = ctx.returns()
    refb = pyoracle.rollout_feedback(pm, pt, start, 0.0, MOCAP, H, 1, 1, 1, nom["times"], nom["states"], nom["actions"], gains, improvement, alpha)
    assert np.array_equal(fail, refb["failure"]) and not fail.any()
    assert close(ret, refb["total_return"], 1e-7)
    tr = ctx.fetch_trajectory(1)
    for name in ("states", "actions", "times", "residual", "costs", "trace"):
        assert close(getattr(tr, name), refb[name][1], 1e-7), name
    ctx.close()


@pytest.mark.parametrize("risk", [0.0, 0.7])
def test_cost_derivatives_at_the_a1s_shape(quad, risk):
    """cost_derivatives_kernel against oracle ocost_derivatives (mjpc/planners/cost_derivatives.cc:112-230) at the shape of
    BASELINE configs[4]: nr = 42 residuals in 9 terms with the task's own norm mix, ndx = 36, nu = 12, T = 36, risk-neutral and
    risk-sensitive. C, D are the ORACLE's finite differences, so both sides contract the same Jacobians: 1e-10 relative."""
    import copy
    t2 = copy.copy(quad); t2.risk = risk
    pm, pt = t2.packed_model(), t2.packed()
    H = 36
    home = quad.model.keyframes["home"]["qpos"]
    rng = np.random.default_rng(11)
    state = np.concatenate([home, 0.3 * rng.normal(size=18)])
    state[7:19] += 0.15 * rng.normal(size=12)
    times = np.arange(4) * (H - 1) * 0.01 / 3
    nodes = np.clip(rng.normal(0, 0.3, (1, 4, 12)), -1, 1)
    nom = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, 1, H, 4, 1, times, nodes, num_threads=1)
    nom = {k: v[0] for k, v in nom.items() if k not in ("total_return", "failure")}
    assert nom["residual"].shape == (H, 42) and pt.struct.num_term == 9
    norms = sorted(set(int(pt.struct.norm[i]) for i in range(9)))
    assert len(norms) >= 3, norms                      # the A1's norm mix (quadratic, L2, smooth-abs ...), not one type
    _, _, Co, Do = pyoracle.transition_fd(pm, pt, nom["states"], nom["times"], nom["actions"], 1e-6, 0, mocap=MOCAP)
    assert Co.shape == (H, 42, 36) and Do.shape == (H, 42, 12)
    ctx = capi.Context(pm, pt, 0, 64)
    got = ctx.cost_derivatives(nom["residual"], Co, Do)
    ref = pyoracle.cost_derivatives(pt, nom["residual"], Co, Do)
    for name, g, o in zip(("cx", "cu", "cxx", "cxu", "cuu"), got, ref):
        scale = 1 + np.abs(o).max()
        assert np.abs(g - o).max() <= 1e-10 * scale, (name, float(np.abs(g - o).max()), scale)
    assert np.abs(got[2]).max() > 0 and np.abs(got[4]).max() > 0
    assert (np.abs(got[3]).max() > 0) == (risk != 0)   # (no term of this task reads both the state and the controls: cxu is the risk transform's cross term)
    ctx.close()


def test_one_ilqg_iteration_on_the_a1_against_the_oracle(quad):
    """BASELINE configs[4]'s iteration, device against oracle, with the ORACLE's finite differences fed to both sides so that the 1 / eps
    amplification of the sweep drops out (ilqg/planner.cc:377-627): cost derivatives -> Riccati pass at the planner's regularisation (state-
    control regularisation, action limits: n = 36, m = 12, T = 36) -> the ten line-search rollouts under the index policy. The gains, the
    improvement and the expected decrease at 1e-8, the rollouts' returns at 1e-7, and the SAME line-search step wins."""
    pm, pt = quad.packed_model(), quad.packed()
    H = 36
    home = quad.model.keyframes["home"]["qpos"]
    rng = np.random.default_rng(21)
    state = np.concatenate([home, 0.2 * rng.normal(size=18)])
    times = np.arange(4) * (H - 1) * 0.01 / 3
    nodes = np.clip(rng.normal(0, 0.2, (1, 4, 12)), -1, 1)
    nom = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, 1, H, 4, 1, times, nodes, num_threads=1)
    nom = {k: v[0] for k, v in nom.items() if k not in ("total_return", "failure")}
    Ao, Bo, Co, Do = pyoracle.transition_fd(pm, pt, nom["states"], nom["times"], nom["actions"], 1e-6, 0, mocap=MOCAP)
    Ao[-1] = 0; Bo[-1] = 0; Do[-1] = 0                       # (the last step has no transition: model_derivatives.cc:88-92)
    limits = np.tile([-1.0, 1.0], (12, 1))
    ctx = capi.Context(pm, pt, 0, 64)
    cd_dev = ctx.cost_derivatives(nom["residual"], Co, Do)
    cd_ref = pyoracle.cost_derivatives(pt, nom["residual"], Co, Do)
    mu, reg_type = 1.0, 1
    dev = ref = None
    for _ in range(8):                                        # the planner's regularisation retries: both sides must agree on success
        dev = ctx.backward_pass(mu, reg_type, 1, Ao, Bo, *cd_dev, nom["actions"], limits)
        ref = pyoracle.riccati(36, 12, H, mu, reg_type, 1, Ao, Bo, *cd_ref, nom["actions"], limits)
        assert dev["ok"] == ref["ok"]
        if ref["ok"]:
            break
        mu *= 1.6
    assert ref["ok"]
    for name in ("K", "du", "Vx", "Vxx"):
        scale = 1 + np.abs(ref[name]).max()
        assert np.abs(dev[name] - ref[name]).max() <= 1e-8 * scale, (name, float(np.abs(dev[name] - ref[name]).max()), scale)
    assert np.allclose(dev["dV"], ref["dV"], rtol=1e-8, atol=1e-10)
    alpha = np.concatenate([np.exp(np.linspace(0, np.log(1e-3), 9)), [0.0]])
    ctx.set_state(state, 0.0, MOCAP)
    ctx.rollout_feedback(H, 0, 0, 1, nom["times"], nom["states"], nom["actions"], dev["K"], dev["du"], alpha)
    ret, fail = ctx.returns()
    refb = pyoracle.rollout_feedback(pm, pt, state, 0.0, MOCAP, H, 0, 0, 1, nom["times"], nom["states"], nom["actions"], ref["K"], ref["du"], alpha)
    assert np.array_equal(np.asarray(fail, bool), np.asarray(refb["failure"], bool))
    ok = ~np.asarray(fail, bool)
    assert ok.any() and close(ret[ok], refb["total_return"][ok], 1e-7)
    best = lambda r, f: min((j for j in range(len(r)) if not f[j]), key=lambda j: (r[j], -j))   # BestRollout: last index wins ties
    assert best(ret, fail) == best(refb["total_return"], refb["failure"])
    assert close(ret[-1], nom["costs"].mean(), 1e-7)          # step 0 reproduces the nominal trajectory
    ctx.close()


def test_ilqg_planner_on_the_quadruped():
    """BASELINE configs[4] in miniature: iLQG on the A1 (T = 36, 10 line-search rollouts, forward differences) -- the
    device sweep (49 perturbed steps per time step), cost derivatives, the MFMA Riccati pass at n = 36, m = 12 and the
    feedback line search all run; the planned return improves; the C++ planner reproduces the Python mirror."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuILQGPlanner, State
    t = load_task("QuadrupedFlat")
    t.transition(0.0)
    H = t.planning_steps()
    assert H == 36
    py = GpuILQGPlanner()
    py.initialize(t.model, t); py.allocate(); py.reset(H)
    st = State(t.model)
    home = t.model.keyframes["home"]["qpos"]
    mp = MOCAP.reshape(2, 7)[:, :3]; mq = MOCAP.reshape(2, 7)[:, 3:]
    st.set(home, np.zeros(18), mocap_pos=mp, mocap_quat=mq, time=0.0)
    py.set_state(st)
    cpp = HostPlanner(load_task("QuadrupedFlat"), kind="ilqg")
    cpp.task_transition(0.0)
    cpp.reset(H)
    cpp.set_state(home, np.zeros(18), 0.0, mocap_pos=mp, mocap_quat=mq)
    returns = []
    for k in range(3):
        py.optimize_policy(H)
        cpp.optimize_policy(H)
        returns.append(py.policy.trajectory.total_return)
        info = cpp.ilqg_info()
        assert info["winner"] == py.winner and info["regularization"] == py.regularization
        ct, cx, cu, cK = cpp.ilqg_policy(H)
        tr = py.policy.trajectory
        assert np.array_equal(cx, tr.states[:H]) and np.array_equal(cu, tr.actions[:H]) and np.array_equal(cK, py.policy.feedback_gain[:H])
    assert np.all(np.isfinite(returns)) and returns[-1] <= returns[0]
    assert py.policy.feedback_gain[:H - 1].any()      # the backward pass produced gains
    # ActionFromPolicy with quaternion StateDiff
    x = np.concatenate([home, np.zeros(18)]); x[0] += 0.01; x[7] += 0.02
    a = np.zeros(12)
    py.action_from_policy(a, x, 0.013)
    np.testing.assert_allclose(cpp.action(0.013, state=x), a, rtol=1e-12, atol=1e-14)


def test_many_constraint_rows_roll_out_like_the_oracle(quad):
    """large exploration noise makes some candidates fall: a fallen A1 has > 64 constraint rows (12 friction-loss rows, joint
    limits, four condim-6 feet and a dozen condim-3 body contacts). MuJoCo grows its arena and rolls such a candidate out;
    the oracle carries a MuJoCo-sized arena (1024 rows / 256 contacts) and the device (no row table: wave_tree.h) must agree --
    the first build failed these rollouts at 64 rows / 16 contacts."""
    home = quad.model.keyframes["home"]["qpos"]
    state = np.concatenate([home, np.zeros(18)])
    pm, pt = quad.packed_model(), quad.packed()
    N, P, H = 16, 4, 100
    rng = np.random.default_rng(0)
    times = np.arange(P) * (H - 1) * 0.01 / (P - 1)
    nodes = np.clip(rng.normal(0, 0.3, (N, P, 12)), -1, 1)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, N, H, P, 1, times, nodes, num_threads=4)
    assert not ref["failure"].any()
    # the scenario does exceed the old caps: replay the oracle's states and count rows
    ph = pyoracle.Physics(pm)
    most_rows = most_contacts = 0
    for c in range(N):
        for s in range(0, H, 5):
            st = ref["states"][c][s]
            ph.set_state(st[:19], st[19:], s * 0.01, MOCAP)
            ph.set_ctrl(ref["actions"][c][s])
            ph.forward()
            most_rows = max(most_rows, int(ph.get("nefc")[0]))
            most_contacts = max(most_contacts, int(ph.get("ncon")[0]))
    assert most_rows > 64, (most_rows, most_contacts)
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.0, MOCAP)
    ctx.rollout_splines(H, 1, times, nodes)
    ret, fail = ctx.returns()
    assert np.array_equal(fail, ref["failure"]), [hex(int(x)) for x in ctx.failure_raw]
    assert close(ret, ref["total_return"], 1e-5)
    ctx.close()


@pytest.mark.parametrize("N,H,P,interp", [(1, 1, 1, 0), (1, 2, 1, 2), (3, 2, 2, 1), (65, 3, 1, 0)])
def test_degenerate_shapes(quad, N, H, P, interp):
    """single candidate, horizon 1 (only the final mj_forward), a single spline node, a batch that is not a multiple of 64"""
    home = quad.model.keyframes["home"]["qpos"]
    state = np.concatenate([home, np.zeros(18)])
    pm, pt = quad.packed_model(), quad.packed()
    rng = np.random.default_rng(N + H)
    times = np.arange(P) * 0.01
    nodes = np.clip(rng.normal(0, 0.2, (N, P, 12)), -1, 1)
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.25, MOCAP)
    ctx.rollout_splines(H, interp, times + 0.25, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, 0.25, MOCAP, N, H, P, interp, times + 0.25, nodes, num_threads=2)
    assert np.array_equal(fail, ref["failure"]) and close(ret, ref["total_return"], 1e-9)
    tr = ctx.fetch_trajectory(N - 1)
    for name in ("states", "actions", "times", "residual", "costs", "trace"):
        assert close(getattr(tr, name), ref[name][N - 1], 1e-9), name
    ctx.close()


def test_maximum_horizon(quad):
    """kMaxTrajectoryHorizon = 512 steps (mjpc/trajectory.h): buffers and the time loop hold; a standing robot under its
    home controls stays finite"""
    home = quad.model.keyframes["home"]["qpos"]
    state = np.concatenate([home, np.zeros(18)])
    pm, pt = quad.packed_model(), quad.packed()
    N, H, P = 2, 512, 3
    times = np.array([0.0, 2.0, 5.11])
    nodes = np.tile(home[7:], (N, P, 1))
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.0, MOCAP)
    ctx.rollout_splines(H, 0, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, N, H, P, 0, times, nodes, num_threads=2)
    assert not fail.any() and np.array_equal(fail, ref["failure"])
    assert close(ret, ref["total_return"], 1e-6)
    tr = ctx.fetch_trajectory(1)
    assert tr.states.shape == (512, 37) and np.isfinite(tr.states).all() and abs(tr.times[-1] - 5.11) < 1e-9
    ctx.close()


@pytest.mark.parametrize("precision,tol", [(64, 1e-7), (32, 2e-3)])
def test_noisy_rollout_matches_the_oracle(quad, precision, tol):
    """Trajectory::NoisyRollout (RobustPlanner): Ornstein-Uhlenbeck xfrc_applied noise from the shared counter-based stream;
    candidates with the SAME spline diverge from each other, and each one tracks the oracle's rollout with the same noise"""
    home = quad.model.keyframes["home"]["qpos"]
    state = np.concatenate([home, np.zeros(18)])
    pm, pt = quad.packed_model(), quad.packed()
    N, P, H = 6, 3, 30
    times = np.arange(P) * 0.15
    nodes = np.tile(home[7:], (N, P, 1))
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, N, H, P, 0, times, nodes, num_threads=4, xfrc_std=0.3, xfrc_rate=0.1,
                                 seed=11, candidate_offset=5)
    assert np.abs(ref["states"][0] - ref["states"][1]).max() > 1e-3      # the noise does something
    ctx = capi.Context(pm, pt, 0, precision)
    ctx.set_state(state, 0.0, MOCAP)
    ctx.rollout_splines_noisy(H, 0, times, nodes, 0.3, 0.1, seed=11, candidate_offset=5)
    ret, fail = ctx.returns()
    assert not fail.any() and close(ret, ref["total_return"], tol)
    for c in (0, N - 1):
        tr = ctx.fetch_trajectory(c)
        assert close(tr.states, ref["states"][c], 30 * tol), float(np.abs(tr.states - ref["states"][c]).max())
    # xfrc_std = 0 is the plain rollout
    ctx.rollout_splines_noisy(H, 0, times, nodes, 0.0, 0.1)
    r0, _ = ctx.returns()
    ctx.rollout_splines(H, 0, times, nodes)
    r1, _ = ctx.returns()
    assert np.array_equal(r0, r1)
    ctx.close()


@pytest.mark.parametrize("name", ["QuadrupedFlat", "HumanoidTrack"])
def test_kinematics_query(name):
    """mjpcx_kinematics: the mjData fields a Task::Transition reads (body / site poses, subtree centre of mass and linear
    velocity) for the state given to mjpcx_set_state, against the oracle's mj_forward"""
    t = load_task(name)
    if name == "QuadrupedFlat":
        t.transition(0.0)
    else:
        t.transition(0.0, mode=3)
    m = t.model
    rng = np.random.default_rng(4)
    q = np.array(m.arrays["qpos0"], float)
    q[:3] += rng.normal(0, 0.2, 3)
    q[3:7] = rng.normal(0, 1, 4); q[3:7] /= np.linalg.norm(q[3:7])
    q[7:] += rng.normal(0, 0.3, m.nq - 7)
    v = rng.normal(0, 1.0, m.nv)
    mocap = np.concatenate([np.concatenate([rng.normal(0, 1, 3), [1, 0, 0, 0]]) for _ in range(m.nmocap)])
    pm, pt = t.packed_model(), t.packed()
    ph = pyoracle.Physics(pm)
    ph.set_state(q, v, 0.3, mocap)
    ph.set_ctrl(np.zeros(m.nu))
    ph.forward()
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(np.concatenate([q, v]), 0.3, mocap)
    k = ctx.kinematics(m.nbody, m.nsite)
    live_b = m.nbody if name == "QuadrupedFlat" else m.nbody - 16   # the humanoid's inert mocap bodies are not on the device
    live_s = m.nsite if name == "QuadrupedFlat" else m.nsite - 16
    for key, width, live in (("xpos", 3, live_b), ("xquat", 4, live_b), ("xmat", 9, live_b), ("xipos", 3, live_b),
                             ("subtree_com", 3, live_b), ("subtree_linvel", 3, live_b), ("site_xpos", 3, live_s)):
        ref = ph.get(key, cap=16384).reshape(-1, width)
        first = 1 if key == "subtree_linvel" else 0   # (the oracle's helper does not define the world body's entry)
        assert close(k[key][first:live], ref[first:live], 1e-12), key
        assert np.all(k[key][live:] == 0)
    ctx.close()


def test_ilqg_kernels_on_random_states():
    """tools/fuzz_ilqg.py: 12 random A1 states -- a nominal trajectory each, feedback rollouts in both policy modes and all three
    representations with random gains (1e-7), forward and centred derivative sweeps (5e-5). 40 such cases: profiles/r03_fuzz_ilqg.log"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_ilqg.py"), "12", "5"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert out.stdout.splitlines()[-1].startswith("12 cases")
