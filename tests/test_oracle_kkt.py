"""Two further derivations of the constrained acceleration the oracle's primal Newton solver returns (oracle/contact.inc
o_constraint_newton), so that the constraint solve the device kernels are compared with does not rest on one piece of code:

 (a) the optimality (KKT) conditions of MuJoCo's convex contact model, evaluated in numpy from the rows alone: with
     y = J qacc - aref and w = y + R f,  M (qacc - qacc_smooth) = J' f  and, per row kind, f in Omega, w in Omega*, f . w = 0
     (friction loss: |f| <= floss with w = 0 inside and w pushing outward at the bounds; limits / frictionless / pyramid rows:
     f >= 0, w >= 0; elliptic contacts: f in the friction cone, w in its dual cone);
 (b) the DUAL problem in the constraint forces solved by projected Gauss-Seidel (contact.inc o_solve_pgs -- the scheme of MuJoCo's
     PGS solver, the one BASELINE.json's north star names), which shares only the rows and the factor of M with the Newton path.

States: the A1 standing, trotting on random controls, and tumbling onto its body geoms (elliptic condim-3 / condim-6 contacts,
friction loss, joint limits); the humanoid walking and collapsing (pyramidal contacts, tendon limits, self collision)."""
import numpy as np
import pytest

from mujoco_mpc_amd.task import load_task
from oracle import pyoracle

FRICTION, LIMIT, NORMAL, ELLIPTIC, CONE_ROW, TENDON, PYRAMID = range(7)
A1_MOCAP = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0])


def _tight(task):
    pm = task.packed_model()
    pm.struct.solver_tolerance = 1e-15   # run Newton to machine precision: the comparison is of the optimum, not of the stopping rule
    pm.struct.solver_iterations = 200
    return pm


def _states(ph, m, q, v, mocap, steps, every, seed, std):
    rng = np.random.default_rng(seed)
    ph.set_state(q, v, 0.0, mocap)
    for s in range(steps):
        ph.set_ctrl(np.clip(rng.normal(0, std, m.nu), -1, 1))
        if s % every == 0:
            ph.forward()
            yield s
        ph.step()


def _kkt(ph, m):
    nv = m.nv
    ne = int(ph.get("nefc")[0])
    if ne == 0:
        assert np.array_equal(ph.get("qacc"), ph.get("qacc_smooth"))
        return set(), 0
    J = ph.get("efc_J", cap=ne * nv + 1).reshape(ne, nv)
    aref, R, f = ph.get("efc_aref"), ph.get("efc_R"), ph.get("efc_force")
    kind, ident, floss = ph.get("efc_type").astype(int), ph.get("efc_id").astype(int), ph.get("efc_floss")
    M = ph.get("M").reshape(nv, nv)
    qacc, qs = ph.get("qacc"), ph.get("qacc_smooth")
    fric = ph.get("contact_friction").reshape(-1, 6)
    con = ph.get("contact", cap=11 * 256).reshape(-1, 11)
    fscale = 1 + np.abs(f).max()
    # stationarity
    assert np.abs(M @ (qacc - qs) - J.T @ f).max() <= 1e-9 * (1 + np.abs(J.T @ f).max())
    y = J @ qacc - aref
    w = y + R * f
    wscale = 1 + np.abs(y).max()
    seen = set()
    r = 0
    while r < ne:
        k = kind[r]
        seen.add(k)
        if k == FRICTION:
            assert abs(f[r]) <= floss[r] * (1 + 1e-12)
            if abs(f[r]) < floss[r] * (1 - 1e-9):
                assert abs(w[r]) <= 1e-9 * wscale
            else:
                assert -np.sign(f[r]) * w[r] >= -1e-9 * wscale     # at a bound the slack pushes outward
        elif k in (LIMIT, NORMAL, TENDON, PYRAMID):
            assert f[r] >= 0 and w[r] >= -1e-9 * wscale and abs(f[r] * w[r]) <= 1e-9 * fscale * wscale
        elif k == ELLIPTIC:
            dim = int(con[ident[r]][9])
            mu = fric[ident[r]][1:dim]
            fc, wc = f[r:r + dim], w[r:r + dim]
            assert fc[0] >= 0 and np.linalg.norm(fc[1:] / mu) <= fc[0] * (1 + 1e-9) + 1e-9 * fscale           # friction cone
            assert wc[0] - np.linalg.norm(wc[1:] * mu) >= -1e-8 * wscale                                      # its dual cone
            assert abs(fc @ wc) <= 1e-8 * fscale * wscale
            r += dim - 1
        else:
            raise AssertionError(k)
        r += 1
    return seen, ne


def _check(ph, m, seen_all):
    seen, ne = _kkt(ph, m)
    seen_all |= seen
    if ne == 0:
        return 0
    qacc, f = ph.get("qacc"), ph.get("efc_force")
    qp, fp, sweeps = ph.solve_pgs()
    assert sweeps < 200000
    assert np.abs(qp - qacc).max() <= 1e-7 * (1 + np.abs(qacc).max()), (np.abs(qp - qacc).max(), sweeps)
    assert np.abs(fp - f).max() <= 1e-7 * (1 + np.abs(f).max())
    return ne


def test_a1_newton_optimum_satisfies_the_kkt_conditions_and_equals_the_pgs_dual():
    t = load_task("QuadrupedFlat")
    t.transition(0.0)
    pm = _tight(t)
    m = pm.struct
    ph = pyoracle.Physics(pm)
    home = t.model.keyframes["home"]["qpos"]
    seen, most = set(), 0
    for _ in _states(ph, m, home, np.zeros(18), A1_MOCAP, 60, 6, 0, 0.4):
        most = max(most, _check(ph, m, seen))
    q = home.copy(); q[2] = 0.45; q[3:7] = np.array([0.9, 0.3, 0.2, 0.1]) / np.linalg.norm([0.9, 0.3, 0.2, 0.1])
    v = np.zeros(18); v[3:6] = [2.0, -1.0, 0.5]
    for _ in _states(ph, m, q, v, A1_MOCAP, 80, 8, 1, 0.3):
        most = max(most, _check(ph, m, seen))
    assert {FRICTION, ELLIPTIC} <= seen and most > 40, (seen, most)


def test_humanoid_newton_optimum_satisfies_the_kkt_conditions_and_equals_the_pgs_dual():
    t = load_task("HumanoidTrack")
    pm = _tight(t)
    m = pm.struct
    ph = pyoracle.Physics(pm)
    seen, most = set(), 0
    for mode, v3, std, seed in ((9, [0.0, 0.0, 0.0], 0.3, 2), (4, [1.5, -1.0, 0.5], 0.8, 3)):
        e = t.transition(0.0, mode=mode)
        mocap = np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(e["mocap_pos"]).reshape(-1, 3)])
        v = np.array(e["qvel"], float).copy(); v[3:6] += v3
        for _ in _states(ph, m, e["qpos"], v, mocap, 80, 8, seed, std):
            most = max(most, _check(ph, m, seen))
    assert PYRAMID in seen and most > 30, (seen, most)
