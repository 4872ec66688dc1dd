"""-m gpu: the limb kernel (csrc/limb_step.h: four lanes per candidate, one per limb of the Humanoid of BASELINE configs[3]) against the CPU
oracle through the C ABI. fp64 (the kernel at the oracle's precision) at 1e-9 (1 + |x|) on every Trajectory buffer over short
horizons and 1e-7 over the config's 64 steps; fp32 -- the precision configs[3] is quoted in and the kernel's default -- at 2e-3 on returns.
A candidate the limb form does not cover is handed to rollout_tree_kernel<Humanoid>: results never depend on which kernel ran."""
import os

import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

pytestmark = pytest.mark.gpu


def mocap7(mpos):
    return np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(mpos).reshape(-1, 3)])


def close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= tol * (1 + np.abs(b)))


@pytest.fixture(scope="module")
def walk():
    t = load_task("HumanoidTrack")
    e = t.transition(0.0, mode=9)
    return t, np.concatenate([e["qpos"], e["qvel"]]), mocap7(e["mocap_pos"])


def limb_context(pm, pt, precision, min_n=0):
    env = {"MJPCX_LIMB_MIN_N": str(min_n)}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = capi.Context(pm, pt, 0, precision)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert ctx.kernel_name.startswith("rollout_limb_kernel"), ctx.kernel_name
    return ctx


def run(walk, N, H, P, interp, seed, precision, tol, std=0.3, time=0.0, node_span=None):
    t, state, mocap = walk
    pm, pt = t.packed_model(), t.packed()
    rng = np.random.default_rng(seed)
    dt = t.model.get_number("agent_timestep", t.model.timestep)
    times = time + np.arange(P) * max((H - 1) * dt / max(P - 1, 1), 1e-3)
    if node_span is not None:   # (first node, last node) in steps after the rollout's start: the spline's two constant ends are exercised
        times = time + dt * np.linspace(node_span[0], node_span[1], P)
    nodes = np.clip(rng.normal(0, std, (N, P, t.model.nu)), -1, 1)
    ctx = limb_context(pm, pt, precision)
    ctx.set_state(state, time, mocap)
    ctx.rollout_splines(H, interp, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, time, mocap, N, H, P, interp, times, nodes, num_threads=8)
    assert np.array_equal(fail, ref["failure"]) and not fail.any()
    st = ctx.quad_stats()
    worst = 0.0
    if precision == 64:
        for c in range(N):
            tr = ctx.fetch_trajectory(c)
            for name in ("states", "actions", "times", "residual", "costs", "trace"):
                g, o = getattr(tr, name), ref[name][c]
                worst = max(worst, float(np.max(np.abs(g - o) / (1 + np.abs(o)))))
                assert close(g, o, tol), (name, c, float(np.max(np.abs(g - o))))
    assert close(ret, ref["total_return"], tol), float(np.max(np.abs(ret - ref["total_return"]) / (1 + np.abs(ref["total_return"]))))
    ctx.close()
    return worst, st


def test_walk_first_steps_fp64(walk):
    worst, st = run(walk, N=8, H=6, P=3, interp=0, seed=1, precision=64, tol=1e-9)
    assert st["handed_on"] == 0


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_walk_sixty_four_steps_fp64(walk, interp):
    worst, st = run(walk, N=32, H=64, P=16, interp=interp, seed=2 + interp, precision=64, tol=1e-7)
    assert st["handed_on"] <= 2   # (noise of std 0.3 on every node: a candidate or two may end up outside the limb form)


@pytest.mark.parametrize("interp", [0, 1, 2])
def test_rollout_longer_than_the_spline_on_both_sides(walk, interp):
    """TimeSpline::Sample before the first node and after the last one returns the end nodes (spline.cc): the rollout starts 6.5 steps before
    the first node and runs 10 steps past the last -- the limb kernel takes the interval index from one lane for the whole wavefront and
    fetches the interval's nodes unconditionally, so both ends and the one-sided cubic slopes next to them are worth their own case"""
    worst, st = run(walk, N=16, H=40, P=6, interp=interp, seed=11 + interp, precision=64, tol=1e-8, std=0.2, node_span=(6.5, 29.0))
    assert st["handed_on"] == 0


def test_walk_fp32_returns(walk):
    worst, st = run(walk, N=64, H=64, P=16, interp=2, seed=7, precision=32, tol=2e-3)
    assert st["handed_on"] <= 4


def test_wild_candidates_are_handed_on_and_still_equal_the_oracle(walk):
    """large noise: some candidates fall (the trunk on the floor, a foot with more contacts than slots, many contacts between moving geoms):
    flagged, rolled out by rollout_tree_kernel<Humanoid>, and the batch still equals the oracle candidate by candidate"""
    worst, st = run(walk, N=64, H=64, P=4, interp=1, seed=3, precision=64, tol=1e-6, std=1.0)
    print("handed on:", st)


def test_device_noise_and_sharding(walk):
    """mjpcx_rollout_noise through the limb kernel: the candidates it draws are the oracle's (Philox keyed on the global index), and the upper
    half of the range as its own launch gives the same returns bit for bit"""
    t, state, mocap = walk
    pm, pt = t.packed_model(), t.packed()
    N, H, P = 256, 32, 8
    dt = t.model.get_number("agent_timestep", t.model.timestep)
    times = np.arange(P) * ((H - 1) * dt / (P - 1))
    nominal = np.clip(np.random.default_rng(5).normal(0, 0.2, (P, t.model.nu)), -1, 1)
    ns = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_SAMPLING, std0=0.1)
    ctx = limb_context(pm, pt, 64)
    ctx.set_state(state, 0.0, mocap)
    ctx.rollout_noise(N, H, 2, times, nominal, ns)
    ret, fail = ctx.returns()
    sample = np.arange(0, N, 8)
    nodes = pyoracle.noise_candidates(pm, ns, P, nominal, sample)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, len(sample), H, P, 2, times, nodes, num_threads=8)
    assert not fail.any() and close(ret[sample], ref["total_return"], 1e-8)
    half = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_SAMPLING, std0=0.1, candidate_offset=N // 2)
    ctx.rollout_noise(N // 2, H, 2, times, nominal, half)
    assert np.array_equal(ctx.returns()[0], ret[N // 2:])
    ctx.close()


@pytest.mark.parametrize("mode", ["cross_entropy", "sampling_two_scales"])
def test_the_other_noise_streams_through_the_limb_kernel(walk, mode):
    """CrossEntropyPlanner::AddNoiseToPolicy (per-parameter variance with the explore / exploit floors, the nominal candidate left clean:
    cross_entropy/planner.cc:351-385) and the sampling planner's second noise scale (a fifth of the candidates: sampling/planner.cc:326-352),
    drawn by the limb kernel itself: the nodes it leaves and the returns are the oracle's"""
    t, state, mocap = walk
    pm, pt = t.packed_model(), t.packed()
    N, H, P = 128, 24, 6
    nu = t.model.nu
    dt = t.model.get_number("agent_timestep", t.model.timestep)
    times = np.arange(P) * ((H - 1) * dt / (P - 1))
    rng = np.random.default_rng(17)
    nominal = np.clip(rng.normal(0, 0.2, (P, nu)), -1, 1)
    if mode == "cross_entropy":
        var = rng.uniform(0.0, 0.05, P * nu) ** 2   # some below the floors, some above
        ns = capi.make_noise_spec(seed=5, iteration=2, mode=capi.NOISE_CROSS_ENTROPY, std0=0.08, std1=0.01, explore_count=N // 4,
                                  nominal_candidate=N - 1, param_variance=var)
    else:
        ns = capi.make_noise_spec(seed=5, iteration=2, mode=capi.NOISE_SAMPLING, std0=0.05, std1=0.2)
    ctx = limb_context(pm, pt, 64)
    ctx.set_state(state, 0.0, mocap)
    ctx.rollout_noise(N, H, 2, times, nominal, ns)
    ret, fail = ctx.returns()
    sample = np.arange(N)
    nodes = pyoracle.noise_candidates(pm, ns, P, nominal, sample)
    for c in (0, 1, N // 4 - 1, N // 4, N - 2, N - 1):
        assert np.max(np.abs(ctx.fetch_spline(c) - nodes[c])) < 1e-13, c   # (device log / cos against libm's: an ulp)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 2, times, nodes, num_threads=8)
    assert np.array_equal(fail, ref["failure"])
    ok = fail == 0
    assert ok.sum() >= N - 4 and close(ret[ok], ref["total_return"][ok], 1e-8)
    ctx.close()


def test_collapse_keeps_tendon_limit_rows_in_the_limb_kernel():
    """From a spinning crouch with near-random controls the body folds: arms and legs touch, the hamstring tendons reach their limits. Candidates
    the limb form covers stay in the limb kernel, the others are handed on -- and the oracle's census of the rollouts says that candidates with
    an active tendon-limit row and with contacts between moving geoms are among those the limb kernel KEPT (its tendon row and its Woodbury
    terms are exercised on the device, not only in the emulator)"""
    from test_gpu_full_size import humanoid_census
    t = load_task("HumanoidTrack")
    e = t.transition(0.0, mode=4)   # Crouch Flip
    v = np.zeros(27)
    v[3:6] = [1.5, -1.0, 0.5]
    state, mocap = np.concatenate([e["qpos"], v]), mocap7(e["mocap_pos"])
    pm, pt = t.packed_model(), t.packed()
    N, H, P = 16, 60, 4
    dt = t.model.get_number("agent_timestep", t.model.timestep)
    times = np.arange(P) * ((H - 1) * dt / (P - 1))
    nodes = np.clip(np.random.default_rng(9).normal(0, 0.8, (N, P, t.model.nu)), -1, 1)
    os.environ["MJPCX_LIMB_NO_FALLBACK"] = "1"   # which candidates the limb kernel kept: the others stay flagged in this context
    try:
        ctx = limb_context(pm, pt, 64)
    finally:
        os.environ.pop("MJPCX_LIMB_NO_FALLBACK", None)
    ctx.set_state(state, 0.0, mocap)
    ctx.rollout_splines(H, 0, times, nodes)
    ret, _ = ctx.returns()
    kept = (np.asarray(ctx.failure_raw) & 0x40000000) == 0
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 0, times, nodes, num_threads=8)
    ok = kept & (ref["failure"] == 0)
    assert ok.sum() >= 4, (int(kept.sum()), int((ref["failure"] == 0).sum()))
    assert close(ret[ok], ref["total_return"][ok], 1e-6)
    idx = np.flatnonzero(ok)
    selfc, tendon, _ = humanoid_census(t, mocap, ref["states"][idx], ref["times"][idx])
    print(f"kept {int(kept.sum())} of {N}; of the {len(idx)} compared: tendon-limit rows in {len(tendon)}, contacts between moving geoms in {len(selfc)}")
    assert len(tendon) >= 1 and len(selfc) >= 1
    ctx.close()
