"""-m gpu: rollout_quad_kernel (mujoco_mpc_amd/csrc/quad_kernel.h: four lanes per candidate, one per leg) through the C ABI against the
oracle, and against the wavefront-per-candidate kernel it hands uncovered candidates to. Tolerances: fp64, 1e-9 (1 + |x|) on every
Trajectory buffer over 100 steps for the candidates the quad kernel rolls out itself (observed 1e-13); candidates it hands on are the
wavefront-per-candidate kernel's and carry that suite's tolerance (1e-6 over these horizons)."""
import os

import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

pytestmark = pytest.mark.gpu
MOCAP = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])


def close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= tol * (1 + np.abs(b)))


@pytest.fixture(scope="module")
def quad():
    t = load_task("QuadrupedFlat")
    t.transition(0.0)
    return t


def context(t, env=None):
    env = dict({"MJPCX_QUAD_MIN_N": "0"}, **(env or {}))  # (batches below 2048 go to the wavefront-per-candidate kernel otherwise)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = capi.Context(t.packed_model(), t.packed(), 0, 64)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ctx.set_state(np.concatenate([t.model.keyframes["home"]["qpos"], np.zeros(18)]), 0.0, MOCAP)
    return ctx


def test_the_quad_kernel_serves_the_a1(quad):
    ctx = context(quad)
    assert "rollout_quad_kernel" in ctx.kernel_name
    ctx.close()
    ctx = context(quad, {"MJPCX_NO_QUAD": "1"})
    assert "rollout_tree_kernel" in ctx.kernel_name
    ctx.close()


def test_a_model_with_unproven_solid_pairs_keeps_the_kernel_that_watches_them(quad):
    """quad_build declines a model in which two solids (box | cylinder) of moving bodies are not plainly proven apart: the quad layout
    cannot watch such a pair, so at ANY batch size the model is served by the wavefront-per-candidate kernel, which does -- the outcome of
    a rollout no longer depends on whether the batch is above the quad kernel's threshold. The A1 with hip cylinders long
    enough to defeat the proof, 4096 candidates: not the quad kernel, and the oracle's returns and failure flags on a sample."""
    m = quad.model
    gt, gb = m.arrays["geom_type"], m.arrays["geom_bodyid"]
    trunk = next(b for b in range(m.nbody) if m.arrays["body_dofnum"][b] == 6)
    pm, pt = quad.packed_model(), quad.packed()
    size = np.ctypeslib.as_array(pm.struct.geom_size, (m.ngeom * 3,)).reshape(-1, 3)
    saved = size.copy()  # (the packed arrays may be the fixture model's own: put them back)
    try:
        for g in range(m.ngeom):
            if gt[g] == 5 and m.arrays["body_parentid"][gb[g]] == trunk:
                size[g, 1] = 0.07
        ctx = capi.Context(pm, pt, 0, 64)
        assert "rollout_quad_kernel" not in ctx.kernel_name
        N, H, P = 4096, 20, 3
        state = np.concatenate([m.keyframes["home"]["qpos"], np.zeros(18)])
        times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
        ns = capi.make_noise_spec(seed=3, iteration=1, mode=capi.NOISE_SAMPLING, std0=0.1)
        ctx.set_state(state, 0.0, MOCAP)
        ctx.rollout_noise(N, H, 0, times, np.zeros((P, 12)), ns)
        ret, fail = ctx.returns()
        ctx.close()
        sample = np.arange(1, N, 128)
        nodes = pyoracle.noise_candidates(pm, ns, P, np.zeros((P, 12)), sample)
        ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, len(sample), H, P, 0, times, nodes, num_threads=8)
    finally:
        size[:] = saved
    assert np.array_equal(ref["failure"] != 0, fail[sample] != 0) and close(ret[sample], ref["total_return"], 1e-8)


@pytest.mark.parametrize("interp,std", [(capi.SPLINE_ZERO, 0.04), (capi.SPLINE_CUBIC, 0.08)])
def test_all_six_buffers_against_the_oracle(quad, interp, std):
    pm, pt = quad.packed_model(), quad.packed()
    N, H, P = 48, 100, 4
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    nominal = np.clip(np.random.default_rng(2).normal(0, 0.05, (P, 12)), -1, 1)
    ns = capi.make_noise_spec(seed=21, iteration=2, mode=capi.NOISE_SAMPLING, std0=std)
    ctx = context(quad)
    ctx.rollout_noise(N, H, interp, times, nominal, ns)
    ret, fail = ctx.returns()
    handed = ctx.quad_stats()
    nodes = pyoracle.noise_candidates(pm, ns, P, nominal, np.arange(N))
    ref = pyoracle.rollout_batch(pm, pt, np.concatenate([quad.model.keyframes["home"]["qpos"], np.zeros(18)]), 0.0, MOCAP, N, H, P, interp, times,
                                 nodes, num_threads=16)
    assert np.array_equal(fail != 0, ref["failure"] != 0)
    tol = 1e-9 if handed["handed_on"] == 0 else 1e-6
    assert close(ret, ref["total_return"], tol)
    for c in range(0, N, 5):
        tr = ctx.fetch_trajectory(c)
        for k in ("states", "actions", "times", "residual", "costs", "trace"):
            assert close(getattr(tr, k), ref[k][c], tol), (c, k)
    ctx.close()


@pytest.mark.parametrize("std", [0.5, 1.0])
def test_tangled_legs(quad, std):
    """large noise: legs cross and tangle (a leg touching two others takes the dense elimination of the leg blocks), calves and feet land on
    other legs' hip CYLINDERS and on their own (csrc/solid_pairs.h; the oracle's contact lists show all of these at std 0.5). A candidate one
    of whose joints is pushed more than 0.2 rad past its (soft) limit leaves the joint box over which the geom pairs the quad layout has no
    place for are proven apart (csrc/pair_cull.h) and is handed to the wavefront-per-candidate kernel, which walks every pair -- the only
    reason anything is handed on here. Either way the returns are the oracle's."""
    pm, pt = quad.packed_model(), quad.packed()
    N, H, P = 64, 100, 3
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    ns = capi.make_noise_spec(seed=5, iteration=1, mode=capi.NOISE_SAMPLING, std0=std)
    nominal = np.zeros((P, 12))
    ctx = context(quad)
    ctx.rollout_noise(N, H, 0, times, nominal, ns)
    ret, fail = ctx.returns()
    st = ctx.quad_stats()
    nodes = pyoracle.noise_candidates(pm, ns, P, nominal, np.arange(N))
    ref = pyoracle.rollout_batch(pm, pt, np.concatenate([quad.model.keyframes["home"]["qpos"], np.zeros(18)]), 0.0, MOCAP, N, H, P, 0, times, nodes, num_threads=16)
    assert not fail.any() and not ref["failure"].any()
    assert close(ret, ref["total_return"], 1e-8), float(np.max(np.abs(ret - ref["total_return"]) / (1 + np.abs(ref["total_return"]))))
    assert 0 < st["handed_on"] <= N and st["handed_on"] == st["out_of_proof_range"], st
    ctx.close()


def test_handed_on_candidates_come_back_from_the_other_kernel(quad):
    """MJPCX_QUAD_CON_CAP=<n> (a test aid: the real limit is the 24 contacts a lane can store): candidates with more than n contacts on one leg are
    rolled out by rollout_tree_kernel; every return equals what a context without the quad kernel computes (to that kernel's tolerance
    against itself: the same code ran)"""
    N, H, P = 64, 100, 3
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    ns = capi.make_noise_spec(seed=5, iteration=1, mode=capi.NOISE_SAMPLING, std0=0.25)
    nominal = np.zeros((P, 12))
    for cap in (3, 4, 5, 6, 8):  # (the smallest cap that splits the batch)
        a = context(quad, {"MJPCX_QUAD_CON_CAP": str(cap)})
        a.rollout_noise(N, H, 0, times, nominal, ns)
        ra, fa = a.returns()
        handed = a.quad_stats()
        if handed["handed_on"] < N:
            break
        a.close()
    assert 0 < handed["handed_on"] < N and 0 < handed["contact_list_full"] <= handed["handed_on"]
    b = context(quad, {"MJPCX_NO_QUAD": "1"})
    b.rollout_noise(N, H, 0, times, nominal, ns)
    rb, fb = b.returns()
    assert np.array_equal(fa, fb)
    # chaotic rollouts at this noise level: the two kernels agree where the dynamics are not yet chaotic; a handed-on candidate is
    # bit-identical (the same kernel rolled it out in both contexts)
    d = np.abs(ra - rb) / (1 + np.abs(rb))
    assert np.sum(d == 0) >= handed["handed_on"] and np.median(d) < 1e-12
    a.close(); b.close()


def test_determinism_and_sharding(quad):
    N, H, P = 256, 40, 3
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    nominal = np.zeros((P, 12))
    ns = capi.make_noise_spec(seed=3, iteration=7, mode=capi.NOISE_SAMPLING, std0=0.04)
    ctx = context(quad)
    ctx.rollout_noise(N, H, 0, times, nominal, ns)
    r1 = ctx.returns()[0].copy()
    ctx.rollout_noise(N, H, 0, times, nominal, ns)
    assert np.array_equal(ctx.returns()[0], r1)
    half = capi.make_noise_spec(seed=3, iteration=7, mode=capi.NOISE_SAMPLING, std0=0.04, candidate_offset=N // 2)
    ctx.rollout_noise(N // 2, H, 0, times, nominal, half)
    assert np.array_equal(ctx.returns()[0], r1[N // 2:])
    ctx.close()


def test_ragged_batches_and_short_horizons(quad):
    """Batches that do not fill a wavefront (16 candidates each) or end inside one, and horizons of one and two steps: a candidate's
    rollout does not depend on its neighbours in the wavefront or on the batch size (bit for bit), and the short horizons agree with the
    oracle (H = 1 is one sensor stage: the return is that step's cost)."""
    pm, pt = quad.packed_model(), quad.packed()
    P = 3
    state = np.concatenate([quad.model.keyframes["home"]["qpos"], np.zeros(18)])
    ctx = context(quad)
    for H in (1, 2, 30):
        times = np.arange(P) * (max(H - 1, 1) * 0.01 / (P - 1))
        nominal = np.clip(np.random.default_rng(5).normal(0, 0.05, (P, 12)), -1, 1)
        ns = capi.make_noise_spec(seed=9, iteration=2, mode=capi.NOISE_SAMPLING, std0=0.05)
        ctx.rollout_noise(64, H, capi.SPLINE_LINEAR, times, nominal, ns)
        full, fail = ctx.returns()
        full = full.copy()
        assert not fail.any() and ctx.quad_stats()["handed_on"] == 0
        nodes = np.stack([ctx.fetch_spline(i) for i in range(64)])
        for N in (1, 3, 17, 49):
            ctx.rollout_noise(N, H, capi.SPLINE_LINEAR, times, nominal, ns)
            r, f = ctx.returns()
            assert r.shape == (N,) and not f.any() and np.array_equal(r, full[:N]), (H, N)
        ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, 8, H, P, capi.SPLINE_LINEAR, times, nodes[:8], num_threads=2)
        assert close(full[:8], ref["total_return"], 1e-9), (H, float(np.abs(full[:8] - ref["total_return"]).max()))
        if H == 1:
            assert close(full[:8], ref["costs"][:, 0], 1e-9)
    ctx.close()


def test_random_states_against_the_oracle():
    """tools/fuzz_quad.py: 60 random plan states (trunk poses and heights, legs far from home, fast initial velocities), horizons, spline
    representations and noise levels, 32 candidates each, against the oracle at 1e-8 (returns) / 1e-7 (states after the horizon);
    candidates the quad kernel hands on are included (they come back from the other kernel). 1050 such cases: profiles/r03_fuzz_quad.log"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_quad.py"), "60", "7"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "60 cases x 32 candidates" in out.stdout.splitlines()[-1]


def test_results_do_not_depend_on_the_candidates_per_wavefront(quad):
    """The launch deals 1, 2, 4, 8 or 16 candidates to a wavefront (small batches spread over every SIMD: quad_kernel.hip); a candidate's
    rollout is its own four lanes' work whatever the others in the wavefront do, so returns, failure flags and a whole trajectory are
    bit-identical across the five shapes -- and with them across world sizes that leave a rank different batch sizes."""
    N, H, P = 200, 60, 3
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    nominal = np.clip(np.random.default_rng(4).normal(0, 0.05, (P, 12)), -1, 1)
    ns = capi.make_noise_spec(seed=5, iteration=1, mode=capi.NOISE_SAMPLING, std0=0.1)
    ref = None
    for cpw in (16, 8, 4, 2, 1):
        ctx = context(quad, {"MJPCX_QUAD_CPW": str(cpw)})
        ctx.rollout_noise(N, H, 0, times, nominal, ns)
        ret, fail = ctx.returns()
        tr = ctx.fetch_trajectory(137)
        got = (ret.copy(), fail.copy(), tr.states.copy(), tr.residual.copy(), tr.costs.copy())
        ctx.close()
        if ref is None:
            ref = got
            assert np.all(np.isfinite(ret)) and not fail.any()
        else:
            for a, b in zip(ref, got):
                assert np.array_equal(a, b), cpw
