"""CPU tests of the limb kernel's step function (mujoco_mpc_amd/csrc/limb_step.h) through its lock-step emulator (tests/limbemu): the SAME
source hipcc compiles for gfx950 -- four lanes per candidate, one per limb of the Humanoid of BASELINE configs[3], cross-lane traffic only
through the quad primitives -- run as four threads per candidate and compared with the oracle. In double: 1e-9 (1 + |x|) on every Trajectory
buffer (1e-12 observed); in float, the precision configs[3] is quoted in: 2e-3 on returns (2e-6 observed). The -m gpu suite
(tests/test_gpu_limb.py) runs the device build of the same source."""
import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle
from tests import limbemu


def mocap7(mpos):
    return np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(mpos).reshape(-1, 3)])


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / (1 + np.abs(b))))


@pytest.fixture(scope="module")
def walk():
    t = load_task("HumanoidTrack")
    e = t.transition(0.0, mode=9)
    return t, np.concatenate([e["qpos"], e["qvel"]]), mocap7(e["mocap_pos"])


def test_the_humanoid_is_in_the_class_the_limb_kernel_covers(walk):
    t, _, _ = walk
    assert limbemu.check(t.packed_model(), t.packed()) == ""


def test_other_models_are_declined_with_a_reason():
    q = load_task("QuadrupedFlat")
    q.transition(0.0)
    assert limbemu.check(q.packed_model(), q.packed()) != ""   # (another residual, elliptic cones, no trunk chain)
    c = load_task("Cartpole")
    assert limbemu.check(c.packed_model(), c.packed()) != ""


def test_forward_pass_against_the_oracle(walk):
    """one mj_forward at the clip's first frame and at a perturbed state with hands on thighs (contacts between moving geoms: the Woodbury
    terms of the Hessian): centre of mass, M, qacc_smooth, the constrained qacc and J' force, and the solver's iteration count"""
    t, state, mocap = walk
    pm, pt = t.packed_model(), t.packed()
    m = t.model
    rng = np.random.default_rng(0)
    ctrl = rng.uniform(-0.5, 0.5, m.nu)
    ph = pyoracle.Physics(pm)
    s2 = state.copy()
    s2[7:28] += rng.normal(0, 0.2, 21); s2[28:] += rng.normal(0, 0.5, 27)
    q = s2[3:7] + rng.normal(0, 0.1, 4); s2[3:7] = q / np.linalg.norm(q)
    seen_cross = 0
    for st in (state, s2):
        ph.set_state(st[:m.nq], st[m.nq:], 0.0, mocap); ph.set_ctrl(ctrl); ph.forward()
        r = limbemu.forward(pm, pt, st, 0.0, mocap, ctrl)
        assert r["flags"] == 0
        assert rel(r["com"], ph.get("subtree_com")[3:6]) < 1e-13
        assert rel(r["M"], ph.get("M").reshape(m.nv, m.nv)) < 1e-12
        assert rel(r["qacc_smooth"], ph.get("qacc_smooth")) < 1e-10
        assert rel(r["qacc"], ph.get("qacc")) < 1e-9 and rel(r["qfrc_constraint"], ph.get("qfrc_constraint")) < 1e-9
        assert r["iters"] == int(ph.get("solver_iter")[0])
        seen_cross += r["nx"]
    assert seen_cross >= 2     # the perturbed state has contacts between moving geoms


@pytest.mark.parametrize("precision,tol_traj,tol_ret", [(64, 1e-9, 1e-9), (32, None, 2e-3)])
def test_rollouts_against_the_oracle(walk, precision, tol_traj, tol_ret):
    """32 candidates of the Predictive-Sampling noise (the emulator draws them itself: Philox keyed on the global index, as the device does),
    the config's 64 steps and 16 cubic spline nodes"""
    t, state, mocap = walk
    pm, pt = t.packed_model(), t.packed()
    m = t.model
    N, H, P = 32, 64, 16
    dt = m.get_number("agent_timestep", m.timestep)
    times = np.arange(P) * ((H - 1) * dt / (P - 1))
    nominal = np.clip(np.random.default_rng(5).normal(0, 0.2, (P, m.nu)), -1, 1)
    ns = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_SAMPLING, std0=0.1)
    nodes = pyoracle.noise_candidates(pm, ns, P, nominal, np.arange(N))
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 2, times, nodes, num_threads=8)
    out = limbemu.rollout(pm, pt, state, 0.0, mocap, N, H, P, 2, times, noise=ns, nominal=nominal, precision=precision)
    assert not out["flags"].any() and not ref["failure"].any()
    assert rel(out["nodes"], nodes) < (1e-15 if precision == 64 else 1e-6)
    if tol_traj is not None:
        for name in ("states", "actions", "times", "residual", "costs", "trace"):
            assert rel(out[name], ref[name]) < tol_traj, name
    assert rel(out["total_return"], ref["total_return"]) < tol_ret
    # the float solver stops at its precision's floor, not on noise: within an iteration per step of the double solver
    if precision == 32:
        ref64 = limbemu.rollout(pm, pt, state, 0.0, mocap, N, H, P, 2, times, noise=ns, nominal=nominal, precision=64)
        assert out["iters"].mean() <= ref64["iters"].mean() + (H - 1)


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_rollout_longer_than_the_spline_on_both_sides(walk, interp):
    """the rollout starts 6.5 steps before the first spline node and runs 10 past the last: the constant ends of TimeSpline::Sample and the
    one-sided cubic slopes next to them, in every interpolation"""
    t, state, mocap = walk
    pm, pt = t.packed_model(), t.packed()
    m = t.model
    N, H, P = 8, 40, 6
    dt = m.get_number("agent_timestep", m.timestep)
    times = dt * np.linspace(6.5, 29.0, P)
    nominal = np.clip(np.random.default_rng(21 + interp).normal(0, 0.2, (P, m.nu)), -1, 1)
    ns = capi.make_noise_spec(seed=5, iteration=1, mode=capi.NOISE_SAMPLING, std0=0.1)
    nodes = pyoracle.noise_candidates(pm, ns, P, nominal, np.arange(N))
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, interp, times, nodes, num_threads=8)
    out = limbemu.rollout(pm, pt, state, 0.0, mocap, N, H, P, interp, times, noise=ns, nominal=nominal, precision=64)
    assert not out["flags"].any() and not ref["failure"].any()
    for name in ("states", "actions", "residual", "costs"):
        assert rel(out[name], ref[name]) < 1e-9, name


def test_wild_candidates_are_flagged_not_approximated(walk):
    """noise of std 1 on four nodes: some candidates leave the limb form (the trunk on the floor, more contacts than slots); the emulator flags
    them (the device hands them to rollout_tree_kernel<Humanoid>) and every unflagged one still equals the oracle"""
    t, state, mocap = walk
    pm, pt = t.packed_model(), t.packed()
    N, H, P = 48, 64, 4
    rng = np.random.default_rng(3)
    dt = t.model.get_number("agent_timestep", t.model.timestep)
    times = np.arange(P) * ((H - 1) * dt / (P - 1))
    nodes = np.clip(rng.normal(0, 1.0, (N, P, t.model.nu)), -1, 1)
    out = limbemu.rollout(pm, pt, state, 0.0, mocap, N, H, P, 1, times, node_values=nodes)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 1, times, nodes, num_threads=8)
    ok = out["flags"] == 0
    assert ok.sum() >= N // 2
    assert rel(out["total_return"][ok], ref["total_return"][ok]) < 1e-6
    assert np.all(out["failure"][~ok] & 0x40000000)
