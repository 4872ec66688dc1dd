#!/usr/bin/env python3
"""Random-state parity sweep of rollout_quad_kernel against the oracle (run on the GPU box): random trunk poses / heights / velocities,
joint angles around the home pose, goals, gait modes of the residual, spline representations and noise levels. Prints the worst
relative error of the returns and of the final states per case, how many candidates were handed to the other kernel, and fails loudly
beyond 1e-8 (returns) / 1e-7 (states after the horizon): contact switching amplifies the 1e-13 per-step agreement -- about one case
in 600 reaches 1.6e-9 on the returns after 76 steps, whichever way the kernel was compiled.
  python tools/fuzz_quad.py [cases] [seed] [tree]      tree: the same sweep on rollout_tree_kernel<A1> (MJPCX_NO_QUAD=1), at 1e-7 / 1e-5"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TREE = len(sys.argv) > 3 and sys.argv[3] == "tree"   # the same sweep on the wavefront-per-candidate kernel of the A1 (rollout_tree_kernel<A1>)
os.environ["MJPCX_NO_QUAD" if TREE else "MJPCX_QUAD_MIN_N"] = "1" if TREE else "0"
import numpy as np
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t = load_task("QuadrupedFlat")
t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
home = t.model.keyframes["home"]["qpos"]
ctx = capi.Context(pm, pt, 0, 64)
assert ctx.kernel_name.startswith("rollout_tree_kernel<A1>" if TREE else "rollout_quad_kernel")
worst_r = worst_s = 0.0
handed = failed = 0
for case in range(cases):
    N, H, P = 32, int(rng.integers(20, 80)), int(rng.integers(2, 6))
    q = home.copy()
    q[0:2] += rng.normal(0, 0.3, 2)
    q[2] += rng.uniform(-0.12, 0.25)                       # from crouched into the floor's margin to dropped from a height
    quat = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.25 if case % 3 else 0.6, 4)
    q[3:7] = quat / np.linalg.norm(quat)
    q[7:] += rng.normal(0, 0.35 if case % 4 else 0.9, 12)   # every fourth case: legs far from home (self-collision, joint limits)
    v = rng.normal(0, 0.5 if case % 5 else 2.5, 18)
    state = np.concatenate([q, v])
    mocap = np.array([rng.normal(0, 1.0), rng.normal(0, 1.0), 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
    interp = int(rng.integers(0, 3))
    times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
    nominal = np.clip(rng.normal(0, 0.2, (P, 12)), -1, 1)
    ns = capi.make_noise_spec(seed=int(rng.integers(1, 1 << 30)), iteration=case, mode=capi.NOISE_SAMPLING, std0=float(rng.choice([0.02, 0.1, 0.4])))
    ctx.set_state(state, 0.01 * case, mocap)
    ctx.rollout_noise(N, H, interp, times, nominal, ns)
    ret, fail = ctx.returns()
    st = {"handed_on": 0} if TREE else ctx.quad_stats()
    handed += st["handed_on"]
    nodes = np.stack([ctx.fetch_spline(i) for i in range(N)])
    ref = pyoracle.rollout_batch(pm, pt, state, 0.01 * case, mocap, N, H, P, interp, times, nodes, num_threads=8)
    ok = ~(np.asarray(fail, bool) | np.asarray(ref["failure"], bool))
    assert np.array_equal(np.asarray(fail, bool), np.asarray(ref["failure"], bool)), (case, fail, ref["failure"])
    failed += int((~ok).sum())
    if ok.any():
        er = np.max(np.abs(ret[ok] - ref["total_return"][ok]) / (1 + np.abs(ref["total_return"][ok])))
        es = 0.0
        for c in np.flatnonzero(ok)[:4]:
            tr = ctx.fetch_trajectory(int(c))
            es = max(es, float(np.max(np.abs(tr.states - ref["states"][c]) / (1 + np.abs(ref["states"][c])))))
        worst_r, worst_s = max(worst_r, er), max(worst_s, es)
        flag = "" if er < (1e-7 if TREE else 1e-8) and es < (1e-5 if TREE else 1e-7) else "   <-- beyond tolerance"
        if st["handed_on"] and not TREE:
            print(f"case {case:3d}: handed on {st}", flush=True)
        if flag or case % 10 == 0:
            print(f"case {case:3d}: H = {H:2d} P = {P} interp {interp} handed on {st['handed_on']:2d} failed {int((~ok).sum()):2d}  returns {er:.2e} states {es:.2e}{flag}", flush=True)
print(f"{cases} cases x 32 candidates: worst returns {worst_r:.3e}, worst states {worst_s:.3e}, handed on {handed}, failed rollouts (both sides) {failed}")
assert worst_r < (1e-7 if TREE else 1e-8) and worst_s < (1e-5 if TREE else 1e-7)   # (the other kernel's sums run in another order: its suite's tolerance)
