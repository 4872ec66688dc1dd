"""-m gpu: (sphere | capsule) x (box | cylinder) pairs between two moving bodies on the device (csrc/solid_pairs.h in the wavefront-per-candidate
kernels: wave_forward.h / wave_tree.h) against the oracle (oracle/contact.inc pair_thin_solid) through the C ABI. Tolerance: fp64, 1e-7 (1 + |x|)
on the states after 80 steps of tumbling contact (the suite's bound for chaotic contact scenes; 1e-11 observed)."""
import os

import numpy as np
import pytest

from mujoco_mpc_amd import capi, mjcf
from mujoco_mpc_amd.task import Task
from oracle import pyoracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def close(a, b, tol):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.all(np.abs(a - b) <= tol * (1 + np.abs(b)))


def scene_state(fm, seed):
    """the thin geoms dropped onto their solids off-centre, spinning and drifting: edge, rim, cap, side and face contacts, geoms rolling off"""
    rng = np.random.default_rng(seed)
    q = fm.arrays["qpos0"].copy()
    v = np.zeros(fm.nv)
    for body in (1, 3):   # the thin geoms (free joints in body order: solid, thin, solid, thin)
        q[7 * body + 0] += rng.uniform(-0.08, 0.08)
        q[7 * body + 1] += rng.uniform(-0.08, 0.08)
        q[7 * body + 2] += rng.uniform(0.0, 0.03)
        quat = rng.normal(size=4)
        quat[0] += 3.0
        q[7 * body + 3:7 * body + 7] = quat / np.linalg.norm(quat)
        v[6 * body:6 * body + 3] = rng.uniform(-0.3, 0.3, 3)
        v[6 * body + 3:6 * body + 6] = rng.uniform(-2, 2, 3)
    return np.concatenate([q, v])


@pytest.mark.parametrize("scene,want", [("a", {(2, 6), (3, 5)}), ("b", {(3, 6), (2, 5)})])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_thin_geoms_on_moving_solids_match_the_oracle(scene, want, seed):
    fm = mjcf.load_xml(os.path.join(HERE, "models", "solids_stack_%s.xml" % scene))
    task = Task(name="scene", residual_id=0, model=fm).reset()
    pm, pt = task.packed_model(), task.packed()
    ctx = capi.Context(pm, pt, 0, 64)
    # (the two solids of the scene -- free bodies 2 m apart -- are the one pair without a narrow phase: reported; the oracle would raise a warning
    # should they come within reach)
    assert ctx.create_warning.startswith("1 collidable geom pair(s)") and "rollout_wave_kernel" in ctx.kernel_name
    state = scene_state(fm, seed)
    H, P, N = 80, 2, 2
    times = np.array([0.0, 1.0])
    nodes = np.zeros((N, P, 1))
    nodes[1] = 0.5
    ctx.set_state(state, 0.0, np.zeros(0))
    ctx.rollout_splines(H, 0, times, nodes)
    ret, fail = ctx.returns()
    ref = pyoracle.rollout_batch(pm, pt, state, 0.0, np.zeros(0), N, H, P, 0, times, nodes, num_threads=2)
    assert not fail.any() and not ref["failure"].any()
    # the scene did what it is for: the oracle saw contacts of all four pair kinds along the way
    ph = pyoracle.Physics(pm)
    gt = fm.arrays["geom_type"]
    kinds = set()
    for t in range(H):
        s = ref["states"][0, t]
        ph.set_state(s[:fm.nq], s[fm.nq:], 0.0, np.zeros(0)); ph.set_ctrl(ref["actions"][0, t]); ph.forward()
        for r in np.array(ph.get("contact")).reshape(-1, 11):
            if fm.arrays["geom_bodyid"][int(r[7])] > 0:
                kinds.add((int(gt[int(r[7])]), int(gt[int(r[8])])))
    assert want <= kinds, kinds
    for c in range(N):
        tr = ctx.fetch_trajectory(c)
        assert close(tr.states, ref["states"][c], 1e-7), (c, float(np.max(np.abs(tr.states - ref["states"][c]))))
    ctx.close()
