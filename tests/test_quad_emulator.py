"""CPU tests of the quad kernel's step function (mujoco_mpc_amd/csrc/quad_step.h) through its lock-step emulator (tests/quademu): the
SAME source hipcc compiles for gfx950 -- four lanes per candidate, one per leg, cross-lane traffic only through the quad primitives -- run
as four threads per candidate and compared with the oracle. Tolerances as in the GPU parity suites (fp64: 1e-9 (1 + |x|) on every
Trajectory buffer); observed 1e-13. The -m gpu suite (tests/test_gpu_quad.py) runs the device build of the same source."""
import numpy as np
import pytest

from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle
from tests import quademu

MOCAP = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])


def close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= tol * (1 + np.abs(b)))


@pytest.fixture(scope="module")
def quad():
    t = load_task("QuadrupedFlat")
    t.transition(0.0)
    return t


def home_state(t):
    return np.concatenate([t.model.keyframes["home"]["qpos"], np.zeros(18)])


def test_the_a1_is_in_the_class_the_quad_kernel_covers(quad):
    assert quademu.check(quad.packed_model(), quad.packed()) == ""


def test_other_models_are_declined_with_a_reason():
    h = load_task("HumanoidTrack")
    h.transition(0.0, mode=9)
    why = quademu.check(h.packed_model(), h.packed())
    assert why != ""  # (the humanoid has tendons, 27 dofs, no legged arrowhead)


def test_two_solids_without_a_plain_proof_decline_the_model(quad):
    """pair_cull.h's contract: a pair outside the quad layout must be proven apart or the model is declined. Two solids (box | cylinder) have
    no narrow phase; the wavefront-per-candidate kernels WATCH them, this layout cannot -- so hip cylinders long enough that the proof over
    the joint ranges no longer holds with the wide pad hand the model to the kernels that watch (until round 5 such pairs were silently
    skipped and the batch size decided whether a rollout could fail on them)."""
    m = quad.model
    gt, gb = m.arrays["geom_type"], m.arrays["geom_bodyid"]
    trunk = next(b for b in range(m.nbody) if m.arrays["body_dofnum"][b] == 6)
    hips = [g for g in range(m.ngeom) if gt[g] == 5 and m.arrays["body_parentid"][gb[g]] == trunk]
    assert len(hips) >= 4
    pm = quad.packed_model()
    size = np.ctypeslib.as_array(pm.struct.geom_size, (m.ngeom * 3,)).reshape(-1, 3)
    saved = size.copy()  # (the packed arrays may be the fixture model's own: put them back)
    try:
        for g in hips:
            size[g, 1] = 0.07  # (half length 4 cm in the model)
        why = quademu.check(pm, quad.packed())
    finally:
        size[:] = saved
    assert "two solids" in why and "proven apart" in why


def test_forward_pass_matches_the_oracle(quad):
    """mj_forward + residual at the home pose (no contact yet) and 30 steps later (feet and calves on the floor, Newton solver active)"""
    pm, pt = quad.packed_model(), quad.packed()
    rng = np.random.default_rng(0)
    ctrl = rng.uniform(-0.3, 0.3, 12)
    ph = pyoracle.Physics(pm)
    ph.set_state(quad.model.keyframes["home"]["qpos"], np.zeros(18), 0.0, MOCAP)
    states = [(home_state(quad), 0.0)]
    for _ in range(30):
        ph.set_ctrl(ctrl)
        ph.step()
    states.append((np.concatenate([ph.get("qpos"), ph.get("qvel")]), 0.3))
    for st, tm in states:
        ph.set_state(st[:19], st[19:], tm, MOCAP)
        ph.set_ctrl(ctrl)
        r = ph.forward_task(pt)
        o = quademu.forward(pm, pt, st, tm, MOCAP, ctrl)
        assert o["flags"] == 0 and o["iters"] == int(ph.get("solver_iter")[0])
        assert close(o["qacc"], ph.get("qacc"), 1e-10) and close(o["qfrc_constraint"], ph.get("qfrc_constraint"), 1e-10)
        assert close(o["M"], ph.get("M").reshape(18, 18), 1e-12)
        assert close(o["residual"], r, 1e-12) and abs(o["cost"] - pyoracle.cost_value(pt, r)) <= 1e-12 * (1 + abs(o["cost"]))


@pytest.mark.parametrize("interp", [capi.SPLINE_ZERO, capi.SPLINE_LINEAR, capi.SPLINE_CUBIC])
def test_rollouts_match_the_oracle(quad, interp):
    """Trajectory::Rollout of noisy candidates: all six buffers and the return (60 steps; standing, stepping, feet sliding)"""
    pm, pt = quad.packed_model(), quad.packed()
    N, H, P = 6, 60, 4
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    nodes = np.clip(np.random.default_rng(3 + interp).normal(0, 0.06, (N, P, 12)), -1, 1)
    ref = pyoracle.rollout_batch(pm, pt, home_state(quad), 0.0, MOCAP, N, H, P, interp, times, nodes, num_threads=4)
    emu = quademu.rollout(pm, pt, home_state(quad), 0.0, MOCAP, N, H, P, interp, times, node_values=nodes)
    ok = emu["flags"] == 0
    assert ok.sum() >= N - 1 and not ref["failure"][ok].any()
    for k in ("states", "actions", "times", "residual", "costs", "trace", "total_return"):
        assert close(emu[k][ok], ref[k][ok], 1e-9), k


def test_candidate_generation_is_the_specified_noise(quad):
    """the lanes draw their own three actuators' nodes: together they are pyoracle.noise_candidates (Philox keyed on the global index)"""
    pm, pt = quad.packed_model(), quad.packed()
    N, H, P = 5, 3, 3
    times = np.arange(P) * 0.01
    nominal = np.clip(np.random.default_rng(1).normal(0, 0.2, (P, 12)), -1, 1)
    ns = capi.make_noise_spec(seed=9, iteration=4, mode=capi.NOISE_SAMPLING, std0=0.1, std1=0.3, candidate_offset=100, nominal_candidate=102)
    emu = quademu.rollout(pm, pt, home_state(quad), 0.0, MOCAP, N, H, P, 0, times, noise=ns, nominal=nominal)
    want = pyoracle.noise_candidates(pm, ns, P, nominal, np.arange(100, 100 + N))
    assert np.array_equal(emu["nodes"], want) and np.array_equal(emu["nodes"][2], nominal)


def _compare_unflagged(quad, seed, sigma, must, tol):
    """candidates the emulator rolled out to the end equal the oracle's; the others may only have left the joint box the bake-time proofs of
    the left-out geom pairs cover (kFlagRange = 64: handed to the wavefront-per-candidate kernel, never computed approximately)"""
    pm, pt = quad.packed_model(), quad.packed()
    N, H, P = 12, 100, 3
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    nodes = np.clip(np.random.default_rng(seed).normal(0, sigma, (N, P, 12)), -1, 1)
    emu = quademu.rollout(pm, pt, home_state(quad), 0.0, MOCAP, N, H, P, 0, times, node_values=nodes)
    assert all(f in (0, 64) for f in emu["flags"])
    ref = pyoracle.rollout_batch(pm, pt, home_state(quad), 0.0, MOCAP, N, H, P, 0, times, nodes, num_threads=4)
    assert not ref["failure"].any()
    for c in must:
        assert emu["flags"][c] == 0 and emu["failure"][c] == 0, c
    for c in range(N):
        if emu["flags"][c]:
            continue
        for k in ("states", "residual", "costs", "trace", "total_return"):
            assert close(emu[k][c], ref[k][c], tol), (k, c)
    return int((emu["flags"] == 0).sum())


def test_legs_touching_each_other(quad):
    """large noise: legs cross -- contacts between two moving geoms (rear foot on front thigh, calf on calf): condim 6 rows that couple two
    lanes' blocks of the Hessian (the lane-pair elimination of arrow_factor); all six buffers still equal the oracle's"""
    assert _compare_unflagged(quad, 5, 0.3, (), 1e-7) >= 8


def test_a_leg_touching_two_others_and_cylinders(quad):
    """larger noise: tangled legs. Found with the oracle's contact lists: candidate 9 of seed 0 and candidate 1 of seed 1 have a leg in contact
    with two others (the leg blocks of the Hessian no longer decouple into pairs: newton_direction_general) and a calf or foot on another
    leg's hip CYLINDER (csrc/solid_pairs.h); candidates 2, 4 and 10 of seed 1 have a foot or calf on their OWN leg's hip cylinder (a contact
    inside one lane, QContact::self). Still the oracle's trajectories (1e-12 observed; these rollouts are chaotic, 1e-8 is the bound)."""
    assert _compare_unflagged(quad, 0, 0.5, (9,), 1e-8) >= 3
    assert _compare_unflagged(quad, 1, 0.5, (1, 2, 4, 10), 1e-8) >= 6


def test_a_joint_beyond_the_proofs_range_hands_the_candidate_on(quad):
    """saturating noise drives joints 0.2 rad and more past their (soft) limits within a few steps. The geom pairs the quad layout has no
    place for (the trunk's boxes and cylinders against the legs, thighs against other legs' hips) are dropped on proofs that cover the joint
    ranges + 0.2 rad (csrc/pair_cull.h): a candidate is flagged at the first step at which a joint is outside that box, not before"""
    pm, pt = quad.packed_model(), quad.packed()
    N, H, P = 12, 40, 3
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    nodes = np.clip(np.random.default_rng(7).normal(0, 1.0, (N, P, 12)), -1, 1)
    emu = quademu.rollout(pm, pt, home_state(quad), 0.0, MOCAP, N, H, P, 0, times, node_values=nodes)
    assert all(f == 64 for f in emu["flags"])
    ref = pyoracle.rollout_batch(pm, pt, home_state(quad), 0.0, MOCAP, N, H, P, 0, times, nodes, num_threads=4)
    rng_ = quad.model.arrays["jnt_range"][1:13]
    for c in range(N):
        q = ref["states"][c, :, 7:19]
        out = np.maximum(rng_[:, 0] - q, q - rng_[:, 1]).max(axis=1)  # per step: how far the worst joint is past its range
        step = (int(emu["failure"][c]) >> 8) & 0xFFFF
        assert out[step] > 0.2 - 1e-9 and (out[:step] <= 0.2 + 1e-9).all(), (c, step)


def test_uncovered_situations_are_flagged_not_computed(quad):
    """what the quad form does not cover -- here a rollout whose accelerations leave the representable range (the oracle reports failure
    for it too) -- is flagged for the wavefront-per-candidate kernel (failure carries kQFallback, the reason and the step), never rolled
    out approximately"""
    pm, pt = quad.packed_model(), quad.packed()
    N, H, P = 2, 20, 3
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    state = home_state(quad)
    state[19 + 3:19 + 6] = 1e12
    emu = quademu.rollout(pm, pt, state, 0.0, MOCAP, N, H, P, 0, times, node_values=np.zeros((N, P, 12)))
    assert emu["flags"].all() and np.all((emu["failure"] & 0x40000000) != 0)
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, N, H, P, 0, times, np.zeros((N, P, 12)), num_threads=2)
    assert ref["failure"].all()


def test_other_modes_of_the_residual(quad):
    """Walk, Scramble, Biped (hand stand) and Flip: the residual through the quad's dealt entries equals the oracle's"""
    for mode in (1, 2, 3, 4):
        t = load_task("QuadrupedFlat")
        t.transition(0.0, mode=mode) if "mode" in t.transition.__code__.co_varnames else t.transition(0.0)
        pm, pt = t.packed_model(), t.packed()
        ph = pyoracle.Physics(pm)
        st = home_state(t)
        ph.set_state(st[:19], st[19:], 0.05, MOCAP)
        ctrl = np.full(12, 0.1)
        ph.set_ctrl(ctrl)
        r = ph.forward_task(pt)
        o = quademu.forward(pm, pt, st, 0.05, MOCAP, ctrl)
        assert o["flags"] == 0 and close(o["residual"], r, 1e-12), mode


def test_random_states_through_the_emulator(quad):
    """The random-state sweep of tools/fuzz_quad.py in miniature, on the CPU: 16 plan states -- trunk poses and heights from dug into the
    floor to dropped, legs far from home every fourth case, fast initial velocities every fifth --, 4 candidates x 25 steps each through
    the quad step function's lock-step emulator against the oracle. Candidates the quad form flags (handed to the other kernel on the
    device) are skipped; the rest agree at 1e-8 on returns (observed 2e-13; all 64 rollouts, none flagged) and the failure flags are the oracle's."""
    pm, pt = quad.packed_model(), quad.packed()
    home = quad.model.keyframes["home"]["qpos"]
    rng = np.random.default_rng(21)
    checked = flagged = 0
    worst = 0.0
    for case in range(16):
        N, H, P = 4, 25, 3
        q = home.copy()
        q[0:2] += rng.normal(0, 0.3, 2)
        q[2] += rng.uniform(-0.12, 0.25)
        quat = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.25 if case % 3 else 0.6, 4)
        q[3:7] = quat / np.linalg.norm(quat)
        q[7:] += rng.normal(0, 0.35 if case % 4 else 0.9, 12)
        state = np.concatenate([q, rng.normal(0, 0.5 if case % 5 else 2.5, 18)])
        mocap = np.array([rng.normal(0, 1.0), rng.normal(0, 1.0), 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
        interp = int(rng.integers(0, 3))
        times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
        nodes = np.clip(rng.normal(0, 0.2, (N, P, 12)), -1, 1)
        ref = pyoracle.rollout_batch(pm, pt, state, 0.01 * case, mocap, N, H, P, interp, times, nodes, num_threads=4)
        emu = quademu.rollout(pm, pt, state, 0.01 * case, mocap, N, H, P, interp, times, node_values=nodes)
        ok = (emu["flags"] == 0) & ~np.asarray(ref["failure"], bool)
        flagged += int((emu["flags"] != 0).sum())
        if ok.any():
            err = np.max(np.abs(emu["total_return"][ok] - ref["total_return"][ok]) / (1 + np.abs(ref["total_return"][ok])))
            worst = max(worst, float(err))
            assert close(emu["states"][ok], ref["states"][ok], 1e-7), case
            checked += int(ok.sum())
    assert worst < 1e-8, worst
    assert checked >= 48 and flagged <= 16, (checked, flagged)


@pytest.mark.parametrize("mode,representation,use_state", [(0, 0, 1), (1, 1, 1), (1, 0, 0), (1, 2, 1)])
def test_feedback_rollouts_match_the_oracle(quad, mode, representation, use_state):
    """The iLQG rollouts on the quad step function (rollout_feedback_quad_kernel on the device): RolloutDiscrete with the index policy and
    Trajectory::Rollout with iLQGPolicy::Action in its three representations, StateDiff on the free joint's quaternion, from an off-nominal
    start -- all buffers against oracle/ilqg.c (1e-9 over 20 steps)."""
    pm, pt = quad.packed_model(), quad.packed()
    H = 20
    rng = np.random.default_rng(12)
    state = home_state(quad)
    times4 = np.arange(4) * (H - 1) * 0.01 / 3
    nodes = np.clip(rng.normal(0, 0.15, (1, 4, 12)), -1, 1)
    nomr = pyoracle.rollout_batch(pm, pt, state, 0.0, MOCAP, 1, H, 4, 1, times4, nodes, num_threads=1)
    nom = {k: v[0] for k, v in nomr.items() if k not in ("total_return", "failure")}
    rng = np.random.default_rng(13)
    gains = 0.05 * rng.normal(size=(H, 12, 36))
    improvement = 0.05 * rng.normal(size=(H, 12))
    alpha = np.concatenate([np.exp(np.linspace(0, np.log(1e-3), 5)), [0.0]])
    start = state.copy()
    start[0:3] += [0.01, -0.005, 0.004]
    q = start[3:7] + [0.0, 0.02, -0.01, 0.015]
    start[3:7] = q / np.linalg.norm(q)
    start[19:] = 0.05 * rng.normal(size=18)
    ref = pyoracle.rollout_feedback(pm, pt, start, 0.0, MOCAP, H, mode, representation, use_state, nom["times"], nom["states"], nom["actions"], gains,
                                    improvement, alpha)
    emu = quademu.rollout_feedback(pm, pt, start, 0.0, MOCAP, len(alpha), H, mode, representation, use_state, nom["times"], nom["states"], nom["actions"],
                                   gains, improvement, alpha)
    assert not emu["flags"].any() and not ref["failure"].any()
    for k in ("states", "actions", "times", "residual", "costs", "trace", "total_return"):
        assert close(emu[k], ref[k], 1e-9), (k, float(np.abs(emu[k] - ref[k]).max()))
    if use_state:
        assert np.ptp(emu["total_return"]) > 1e-7


def test_elimination_plan_on_every_leg_contact_graph():
    """quad_step.h make_plan over all 64 graphs on the four legs: a forest gets an order in which every edge is eliminated exactly once, each
    time from a leg that is a LEAF at that moment into the leg on the other end, at most one elimination per leg, the lower leg of an isolated
    pair; a graph with a cycle is reported cyclic (the kernel hands such a candidate on, kFlagPair)"""
    import itertools
    from tests import quademu
    edges_all = [(a, b) for a in range(4) for b in range(a + 1, 4)]
    seen_cyclic = seen_forest = 0
    for bits in range(64):
        edges = {e for i, e in enumerate(edges_all) if bits >> i & 1}
        masks = [0] * 4
        for a, b in edges:
            masks[a] |= 1 << (a ^ b); masks[b] |= 1 << (a ^ b)
        # a cycle: union-find over the edges
        root = list(range(4))

        def find(k):
            while root[k] != k:
                k = root[k]
            return k
        has_cycle = False
        for a, b in sorted(edges):
            ra, rb = find(a), find(b)
            if ra == rb:
                has_cycle = True
            root[ra] = rb
        nslots, x, eslot, cyclic = quademu.plan(masks)
        assert cyclic == has_cycle, (sorted(edges), cyclic)
        if cyclic:
            seen_cyclic += 1
            continue
        seen_forest += 1
        left = set(edges)
        assert nslots <= 3 and all(0 < x[s] < 4 for s in range(nslots)) and all(-1 <= s < nslots for s in eslot)
        for s in range(nslots):
            gone = set()
            for k in range(4):
                if eslot[k] != s:
                    continue
                e = (min(k, k ^ x[s]), max(k, k ^ x[s]))
                assert e in left and e not in gone, (sorted(edges), s, k)
                assert sum(1 for f in left if k in f) == 1, (sorted(edges), s, k)       # a leaf when its slot comes
                if sum(1 for f in left if (k ^ x[s]) in f) == 1:                          # an isolated pair: the lower leg eliminates
                    assert k < (k ^ x[s])
                assert eslot[k ^ x[s]] != s                                                # the receiver is not eliminated in the same slot
                gone.add(e)
            assert gone, (sorted(edges), s)
            left -= gone
        assert not left, (sorted(edges), sorted(left))
    assert seen_forest == 38 and seen_cyclic == 26   # the labelled forests on four vertices (OEIS A001858: 38)
