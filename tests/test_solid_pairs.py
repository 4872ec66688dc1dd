"""The narrow phase of (sphere | capsule) x (box | cylinder) pairs between moving bodies (mjc_SphereBox / mjc_CapsuleBox / mjc_SphereCylinder /
MuJoCo's convex collider for capsule-cylinder; the A1's trunk and hip solids against its leg geoms): the oracle's C restatement
(oracle/contact.inc thin_vs_solid) against a brute-force distance, against the device's form of the same arithmetic (csrc/solid_pairs.h,
built for the CPU), and through the physics (resting force = weight, the normal's direction). CPU only."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle
from tests import solidpairs


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))


def oracle_tvs(cyl, size, p, a, h, r):
    L = pyoracle.lib()
    L.thin_vs_solid.restype = C.c_double
    L.thin_vs_solid.argtypes = [C.c_int] + [C.POINTER(C.c_double)] * 3 + [C.c_double, C.c_double] + [C.POINTER(C.c_double)] * 2
    n, c = np.zeros(3), np.zeros(3)
    size = np.ascontiguousarray(size, float); p = np.ascontiguousarray(p, float); a = np.ascontiguousarray(a, float)
    d = L.thin_vs_solid(int(cyl), _d(size), _d(p), _d(a), float(h), float(r), _d(n), _d(c))
    return d, n, c


def device_tvs(cyl, size, p, a, h, r):
    n, c = np.zeros(3), np.zeros(3)
    size = np.ascontiguousarray(size, float); p = np.ascontiguousarray(p, float); a = np.ascontiguousarray(a, float)
    d = solidpairs.lib().sp_thin_vs_solid(int(cyl), _d(size), _d(p), _d(a), float(h), float(r), _d(n), _d(c))
    return d, n, c


def point_to_solid(pts, cyl, size):
    if not cyl:
        return np.linalg.norm(np.maximum(np.abs(pts) - size[:3], 0.0), axis=-1)
    rho = np.hypot(pts[..., 0], pts[..., 1])
    return np.hypot(np.maximum(rho - size[0], 0.0), np.maximum(np.abs(pts[..., 2]) - size[1], 0.0))


def brute(cyl, size, p, a, h, r):
    """min over the axis of the point-to-solid distance by nested grids (the function is convex along the axis)"""
    lo, hi = -h, h
    for _ in range(8):
        t = np.linspace(lo, hi, 201)
        f = point_to_solid(p[None, :] + t[:, None] * a[None, :], cyl, size)
        i = int(np.argmin(f))
        lo, hi = t[max(i - 1, 0)], t[min(i + 1, 200)]
    return float(f.min()) - r


def random_case(rng, cyl, sphere):
    size = np.array([rng.uniform(0.02, 0.15), rng.uniform(0.02, 0.15), rng.uniform(0.02, 0.15)])
    a = rng.normal(size=3); a /= np.linalg.norm(a)
    if rng.random() < 0.15:  # axis along a coordinate axis of the solid: the degenerate slopes
        a = np.eye(3)[rng.integers(3)] * rng.choice([-1.0, 1.0])
    p = rng.normal(size=3) * 0.2
    h = 0.0 if sphere else rng.uniform(0.02, 0.2)
    r = rng.uniform(0.005, 0.03)
    return size, p, a, h, r


@pytest.mark.parametrize("cyl", [0, 1])
@pytest.mark.parametrize("sphere", [0, 1])
def test_distance_is_the_minimum_over_the_axis(cyl, sphere):
    rng = np.random.default_rng(10 * cyl + sphere)
    separated = 0
    for _ in range(600):
        size, p, a, h, r = random_case(rng, cyl, sphere)
        d, n, c = oracle_tvs(cyl, size, p, a, h, r)
        assert abs(np.linalg.norm(n) - 1) < 1e-12
        t = float((c - p) @ a)
        assert abs(t) <= h + 1e-12 and np.allclose(p + t * a, c, atol=1e-12)
        D = point_to_solid(c, cyl, size)
        if D > 0:  # axis point outside: the distance is the true minimum, the normal points at the closest point of the solid
            separated += 1
            assert abs(d - (D - r)) < 1e-12
            assert abs(d - brute(cyl, size, p, a, h, r)) < 1e-9
            q = c + n * D
            assert point_to_solid(q, cyl, size) < 1e-12
        else:      # inside: out through the nearest face, deeper than the radius
            assert d <= -r + 1e-12
    assert separated > 300


@pytest.mark.parametrize("cyl", [0, 1])
def test_device_form_is_the_same_arithmetic(cyl):
    rng = np.random.default_rng(7 + cyl)
    worst = 0.0
    for i in range(2000):
        size, p, a, h, r = random_case(rng, cyl, i % 3 == 0)
        d0, n0, c0 = oracle_tvs(cyl, size, p, a, h, r)
        d1, n1, c1 = device_tvs(cyl, size, p, a, h, r)
        worst = max(worst, abs(d0 - d1), np.abs(n0 - n1).max(), np.abs(c0 - c1).max())
    assert worst == 0.0  # (both built without contraction: bit-identical)


def test_fp32_form_agrees_to_single_precision():
    rng = np.random.default_rng(3)
    L = solidpairs.lib()
    f = lambda x: np.ascontiguousarray(x, np.float32).ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    for i in range(500):
        cyl = i & 1
        size, p, a, h, r = random_case(rng, cyl, i % 3 == 0)
        d0, n0, c0 = oracle_tvs(cyl, size, p, a, h, r)
        if abs(d0 + r) < 1e-3 or point_to_solid(c0, cyl, size) == 0:
            continue  # (at the surface or inside: the branch may differ in single precision)
        n, c = np.zeros(3, np.float32), np.zeros(3, np.float32)
        s32, p32, a32 = np.float32(size), np.float32(p), np.float32(a)
        d1 = L.sp_thin_vs_solid_f32(cyl, f(s32), f(p32), f(a32), np.float32(h), np.float32(r), n.ctypes.data_as(C.POINTER(C.c_float)), c.ctypes.data_as(C.POINTER(C.c_float)))
        assert abs(d0 - d1) < 2e-5


def test_known_configurations():
    # sphere above the cap of a cylinder, beside its side, off its rim
    R, H, r = 0.04, 0.04, 0.02
    size = np.array([R, H, 0.0])
    z = np.array([0.0, 0, 1])
    d, n, c = oracle_tvs(1, size, np.array([0.01, 0.0, 0.1]), z, 0.0, r)
    assert abs(d - (0.1 - H - r)) < 1e-15 and np.allclose(n, [0, 0, -1])
    d, n, c = oracle_tvs(1, size, np.array([0.1, 0.0, 0.01]), z, 0.0, r)
    assert abs(d - (0.1 - R - r)) < 1e-15 and np.allclose(n, [-1, 0, 0])
    d, n, c = oracle_tvs(1, size, np.array([0.07, 0.0, 0.08]), z, 0.0, r)
    assert abs(d - (np.hypot(0.03, 0.04) - r)) < 1e-15 and np.allclose(n, [-0.6, 0, -0.8])
    # capsule lying across a box edge: the nearest axis point is where the axis passes the edge
    box = np.array([0.1, 0.1, 0.1])
    a = np.array([0.0, 1.0, 0.0])
    d, n, c = oracle_tvs(0, box, np.array([0.15, 0.3, 0.15]), a, 0.5, 0.01)
    assert abs(d - (np.hypot(0.05, 0.05) - 0.01)) < 1e-15 and abs(c[1]) <= 0.1 + 1e-12
    # capsule pointing at a face: its end is the nearest point
    d, n, c = oracle_tvs(0, box, np.array([0.4, 0.02, -0.03]), np.array([1.0, 0, 0]), 0.2, 0.01)
    assert abs(d - (0.4 - 0.2 - 0.1 - 0.01)) < 1e-15 and np.allclose(n, [-1, 0, 0]) and np.allclose(c, [0.2, 0.02, -0.03])
    # capsule axis through the box: the middle of the crossing, out through the nearest face
    d, n, c = oracle_tvs(0, box, np.array([0.0, 0.0, 0.08]), np.array([1.0, 0, 0]), 0.5, 0.01)
    assert np.allclose(c, [0, 0, 0.08]) and np.allclose(n, [0, 0, -1]) and abs(d - (-0.02 - 0.01)) < 1e-15


def test_axis_parallel_to_a_face_takes_the_middle_of_the_nearest_interval():
    box = np.array([0.1, 0.1, 0.1])
    d, n, c = oracle_tvs(0, box, np.array([0.03, 0.02, 0.2]), np.array([1.0, 0, 0]), 0.5, 0.01)
    assert np.allclose(c, [0.0, 0.02, 0.2], atol=1e-15) and np.allclose(n, [0, 0, -1]) and abs(d - 0.09) < 1e-15
    # a short capsule lying wholly above a cylinder's cap: its own middle
    d, n, c = oracle_tvs(1, np.array([0.1, 0.1, 0.0]), np.array([0.01, 0.0, 0.13]), np.array([1.0, 0, 0]), 0.08, 0.03)
    assert np.allclose(c, [0.01, 0.0, 0.13], atol=1e-15) and np.allclose(n, [0, 0, -1]) and abs(d) < 1e-15


@pytest.mark.parametrize("scene", ["a", "b"])
def test_thin_geoms_rest_on_moving_solids_and_carry_their_weight(scene):
    """two stacks per scene (a: ball on box, capsule on cylinder; b: capsule on box, ball on cylinder; all bodies free): each comes to rest, the
    contact between the two moving geoms carries the upper body's weight along -z (from the thin geom, geom1, to the solid, geom2), and the
    solids' contacts with the floor carry both"""
    import os
    from mujoco_mpc_amd import mjcf
    from mujoco_mpc_amd.task import PackedModel
    fm = mjcf.load_xml(os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "solids_stack_%s.xml" % scene))
    ph = pyoracle.Physics(PackedModel(fm))
    ph.set_state(fm.arrays["qpos0"].copy(), np.zeros(fm.nv))
    ph.set_ctrl(np.zeros(fm.nu))
    for _ in range(2500):
        ph.step()
    assert ph.warning() == 0 and np.abs(ph.get("qvel")).max() < 1e-8
    ph.forward()
    con = ph.get("contact").reshape(-1, 11)
    f = ph.get("efc_force")
    gt = fm.arrays["geom_type"]
    gb = fm.arrays["geom_bodyid"]
    pairs = [r for r in con if gb[int(r[7])] > 0 and gb[int(r[8])] > 0]
    assert len(pairs) == 2
    for r in pairs:
        g1, g2 = int(r[7]), int(r[8])
        assert gt[g1] in (2, 3) and gt[g2] in (5, 6)          # MuJoCo's order: the lower geom type is geom1
        assert np.allclose(r[4:7], [0, 0, -1], atol=1e-9)      # normal: geom1 -> geom2
        assert -1e-3 < r[0] < 0
        assert abs(f[int(r[10])] - 2 * 9.81) < 1e-5            # the upper body's weight
    floor = np.array([f[int(r[10])] for r in con if gb[int(r[7])] == 0])
    assert abs(floor.sum() - 2 * 3 * 9.81) < 1e-4


def test_pairs_proven_apart_are_apart_on_random_joint_samples():
    """csrc/pair_cull.h's proofs on the A1, checked the other way round: 2500 random joint configurations inside the box a proof covers
    (every hinge up to 0.2 rad past its range; the knee's fold side 0.1 rad for the ten pairs whose proof needed that) -- none of the 165
    pairs proven apart (146 + 10 thin-solid, 9 cylinder-cylinder) comes within its margin; the smallest distance seen stays positive.
    Distances from the oracle's geom poses and narrow phase (two cylinders: one taken as its enclosing capsule, as the proof does)."""
    import ctypes as C
    from mujoco_mpc_amd.task import load_task
    quad = load_task("QuadrupedFlat")
    quad.transition(0.0)
    pm = quad.packed_model()
    a = quad.model.arrays
    out = (C.c_int * (6 * 1024))()
    n = solidpairs.lib().sp_moving_pairs(C.cast(pm.ptr, C.c_void_p), out, 1024)
    rows = [tuple(out[6 * i:6 * i + 6]) for i in range(n)]
    proven = [r for r in rows if r[3]]
    assert len(proven) == 165
    ph = pyoracle.Physics(pm)
    rng = np.random.default_rng(0)
    lo, hi = a["jnt_range"][1:13, 0], a["jnt_range"][1:13, 1]
    ng = quad.model.scalars["ngeom"]
    mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
    worst = np.inf
    for it in range(2500):
        pad_lo, pad_hi = np.full(12, 0.2), np.full(12, 0.2)
        pad_lo[2::3] = 0.1   # the knees' fold side: the tight pad of the same-leg proofs (valid for every other proof too)
        q = rng.uniform(lo - pad_lo, hi + pad_hi)
        if it % 4 == 0:      # corners of the box are where proofs fail first
            q = np.where(rng.random(12) < 0.5, lo - pad_lo, hi + pad_hi)
        qpos = np.concatenate([[0, 0, 0.5, 1, 0, 0, 0], q])
        ph.set_state(qpos, np.zeros(18), 0.0, mocap)
        ph.forward()
        gx = np.array(ph.get("geom_xpos", 3 * ng)).reshape(ng, 3)
        gm = np.array(ph.get("geom_xmat", 9 * ng)).reshape(ng, 3, 3)
        for g1, g2, kind, apart, tj, ts in proven:
            t1, t2 = int(a["geom_type"][g1]), int(a["geom_type"][g2])
            s1 = a["geom_size"][g1]
            h, r = (s1[1] if t1 in (3, 5) else 0.0), s1[0]
            R2 = gm[g2]
            pl = R2.T @ (gx[g1] - gx[g2])
            al = R2.T @ gm[g1][:, 2]
            d, _, _ = oracle_tvs(t2 == 5, a["geom_size"][g2], pl, al, h, r)
            worst = min(worst, d)
            assert d > 0.001, (g1, g2, d, it)
    assert worst > 0.001


@pytest.mark.parametrize("rng_,provable", [(0.3, True), (0.9, False)])
def test_the_proof_depends_on_the_joint_ranges(rng_, provable):
    """tests/models/two_arms.xml: a box and a cylinder at the ends of two arms 0.6 m apart on a base that turns freely (an unlimited hinge ABOVE
    both bodies: not between them, so it does not enter the proof). With the arms' hinges limited to +-0.3 rad the ends stay 0.26 m apart
    even 0.2 rad past the limits: proven; with +-0.9 rad they can meet: no proof, and a sampled configuration inside the ranges has them in touch."""
    import ctypes as C
    import os
    from mujoco_mpc_amd import mjcf
    from mujoco_mpc_amd.task import PackedModel
    fm = mjcf.load_xml(os.path.join(os.path.dirname(os.path.abspath(__file__)), "models", "two_arms.xml"))
    fm.arrays["jnt_range"][1] = [-rng_, rng_]
    fm.arrays["jnt_range"][2] = [-rng_, rng_]
    pm = PackedModel(fm)
    out = (C.c_int * (6 * 16))()
    n = solidpairs.lib().sp_moving_pairs(C.cast(pm.ptr, C.c_void_p), out, 16)
    rows = [tuple(out[6 * i:6 * i + 6]) for i in range(n)]
    assert len(rows) == 1 and rows[0][2] == 2          # one pair: (cylinder, box) = two solids
    g1, g2 = rows[0][0], rows[0][1]
    assert fm.arrays["geom_type"][g1] == 5 and fm.arrays["geom_type"][g2] == 6
    assert bool(rows[0][3]) == provable
    # the other way round: distances of the cylinder's enclosing capsule to the box over the joint box
    ph = pyoracle.Physics(pm)
    best = np.inf
    for ql in np.linspace(-rng_, rng_, 25):
        for qr in np.linspace(-rng_, rng_, 25):
            ph.set_state(np.array([0.7, ql, qr]), np.zeros(3))
            ph.forward()
            gx = np.array(ph.get("geom_xpos", 3 * fm.scalars["ngeom"])).reshape(-1, 3)
            gm = np.array(ph.get("geom_xmat", 9 * fm.scalars["ngeom"])).reshape(-1, 3, 3)
            s1 = fm.arrays["geom_size"][g1]
            d, _, _ = oracle_tvs(0, fm.arrays["geom_size"][g2], gm[g2].T @ (gx[g1] - gx[g2]), gm[g2].T @ gm[g1][:, 2], s1[1], s1[0])
            best = min(best, d)
    assert (best > 0.1) if provable else (best < 0.0)
