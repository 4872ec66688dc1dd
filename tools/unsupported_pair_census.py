#!/usr/bin/env python3
"""HISTORICAL (round 4, first session): the pairs this tool counts -- a box or a cylinder against a sphere | capsule between two moving bodies -- are collided
since the round's second session (csrc/solid_pairs.h, oracle/contact.inc pair_thin_solid); it measured what leaving them out was worth.

How often do the geom pairs the kernels (and the oracle) do NOT collide come within their margin on the north-star batch?

`mjpcx_create` leaves out every pair of geoms on two moving bodies in which a geom is neither a sphere nor a capsule (the A1: the trunk's
boxes and cylinders and the hips' cylinders against thigh / calf / foot geoms; csrc/wave_model.h, oracle/contact.inc bake_pairs) and says
so in `mjpcx_create_error()` / the bench line's `config.physics_not_reproduced`. This tool measures what that omission is worth on the
workload the headline is quoted on: it rolls a sample of the north-star candidates with the CPU oracle (the planner's own noise around a
nominal it improves over a few plan steps), recomputes every geom's world pose at every step, and evaluates the EXACT distance of every
left-out pair (sphere | capsule against a box or a solid cylinder: the point-to-solid distance is closed form and convex, so the minimum
along the capsule's segment is found by ternary search; cylinder against cylinder: one of the two is widened to its enclosing capsule,
which can only over-count). A candidate is `affected` from the first step at which some left-out pair is within the pair's margin:
before that step its physics is exactly what MuJoCo's pair filter + colliders would produce, after it a contact force is missing.

CPU only (test infrastructure: it drives oracle/); prints one JSON object.
Usage: python tools/unsupported_pair_census.py [N] [plan steps] [advance]   (advance = 1: the plant follows each plan's winner for one agent
step, as in the agent loop; default 0: the state stays at the home keyframe while the nominal improves, as in bench.py's timed loop)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mujoco_mpc_amd import capi  # noqa: E402
from mujoco_mpc_amd.task import load_task  # noqa: E402
from oracle import pyoracle  # noqa: E402

PLANE, SPHERE, CAPSULE, CYLINDER, BOX = 0, 2, 3, 5, 6   # mjtGeom (include/mjpcx.h MJPCX_GEOM_*)
TYPE_NAME = {PLANE: "plane", SPHERE: "sphere", CAPSULE: "capsule", CYLINDER: "cylinder", BOX: "box"}


def left_out_pairs(m):
    """MuJoCo's body-pair filters as csrc/wave_model.h:247-277 applies them; returns the pairs that pass and are not sphere | capsule both"""
    a_ = m.arrays
    nb, ng = m.scalars["nbody"], m.scalars["ngeom"]
    moving = np.zeros(nb, bool)
    for b in range(1, nb):
        moving[b] = moving[a_["body_parentid"][b]] or a_["body_dofnum"][b] > 0
    weld, parent = a_["body_weldid"], a_["body_parentid"]
    excl = set(int(s) for s in a_["exclude_signature"]) if m.scalars["nexclude"] else set()
    out = []
    for a in range(ng):
        for b in range(a + 1, ng):
            b1, b2 = int(a_["geom_bodyid"][a]), int(a_["geom_bodyid"][b])
            if not (moving[b1] and moving[b2]):
                continue
            if not ((a_["geom_contype"][a] & a_["geom_conaffinity"][b]) or (a_["geom_contype"][b] & a_["geom_conaffinity"][a])):
                continue
            w1, w2 = int(weld[b1]), int(weld[b2])
            if w1 == w2:
                continue
            pw1, pw2 = int(weld[parent[w1]]), int(weld[parent[w2]])
            if w1 != 0 and w2 != 0 and (w1 == pw2 or w2 == pw1):
                continue
            if ((min(b1, b2) << 16) + max(b1, b2)) in excl:
                continue
            ta, tb = int(a_["geom_type"][a]), int(a_["geom_type"][b])
            if ta in (SPHERE, CAPSULE) and tb in (SPHERE, CAPSULE):
                continue
            out.append((a, b))
    return out


def point_to_solid(p, typ, size):
    """distance of points p (..., 3, in the solid's frame) to a box (half sizes) or a solid cylinder (radius, half length along z)"""
    if typ == BOX:
        return np.linalg.norm(np.maximum(np.abs(p) - size[:3], 0.0), axis=-1)
    rho = np.hypot(p[..., 0], p[..., 1])
    return np.hypot(np.maximum(rho - size[0], 0.0), np.maximum(np.abs(p[..., 2]) - size[1], 0.0))


def pair_distance(m, ga, gb, xpos, xmat):
    """exact distance (>= 0; 0: touching or inside) between geoms ga and gb for every (candidate, step): xpos (..., ng, 3), xmat (..., ng, 3, 3)"""
    a_ = m.arrays
    ta, tb = int(a_["geom_type"][ga]), int(a_["geom_type"][gb])
    if ta in (BOX, CYLINDER) and tb in (SPHERE, CAPSULE):
        ga, gb, ta, tb = gb, ga, tb, ta
    sa, sb = a_["geom_size"][ga], a_["geom_size"][gb]
    # the thin geom: a segment (sphere: a point; a cylinder standing in for the thin side: its enclosing capsule) of radius ra
    ra = sa[0]
    half = 0.0 if ta == SPHERE else sa[1]
    axis = xmat[..., ga, :, 2]
    c = xpos[..., ga, :]
    # into the solid's frame
    R = xmat[..., gb, :, :]
    to_local = lambda p: np.einsum("...ji,...j->...i", R, p - xpos[..., gb, :])  # noqa: E731  (R' (p - centre))
    lo, hi = np.full(c.shape[:-1], -1.0), np.full(c.shape[:-1], 1.0)
    f = lambda t: point_to_solid(to_local(c + (t * half)[..., None] * axis), tb, sb)  # noqa: E731
    for _ in range(40 if half > 0 else 1):
        t1, t2 = lo + (hi - lo) / 3, hi - (hi - lo) / 3
        left = f(t1) <= f(t2)
        hi = np.where(left, t2, hi)
        lo = np.where(left, lo, t1)
    return np.maximum(f(0.5 * (lo + hi)) - ra, 0.0)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    plans = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    advance = len(sys.argv) > 3 and sys.argv[3] == "1"
    H = 100
    task = load_task("QuadrupedFlat")
    task.transition(0.0)
    m = task.model
    pm, pt = task.packed_model(), task.packed()
    pairs = left_out_pairs(m)
    a_ = m.arrays
    by_type = {}
    for a, b in pairs:
        k = "-".join(sorted((TYPE_NAME[int(a_["geom_type"][a])], TYPE_NAME[int(a_["geom_type"][b])])))
        by_type[k] = by_type.get(k, 0) + 1
    margin = {p: max(a_["geom_margin"][p[0]], a_["geom_margin"][p[1]]) for p in pairs}

    home = m.keyframes["home"]
    state = np.concatenate([np.asarray(home["qpos"], float), np.zeros(m.nv)])
    ids = sorted((b for b in range(m.scalars["nbody"]) if a_["body_mocapid"][b] >= 0), key=lambda b: a_["body_mocapid"][b])
    mocap = np.hstack([np.array([a_["body_pos"][b] for b in ids]), np.array([a_["body_quat"][b] for b in ids])]).reshape(-1)
    P = int(m.get_number("sampling_spline_points", 3))
    dt = m.get_number("agent_timestep", m.timestep)
    times = np.array([k * (H - 1) * dt / (P - 1) for k in range(P)])
    sigma = m.get_number("sampling_exploration", 0.5)
    nominal = np.zeros((P, m.nu))
    threads = os.cpu_count() or 1
    ph = pyoracle.Physics(pm)
    report = {"candidates_sampled": N, "horizon": H, "plan_steps": plans, "state_advances": advance, "left_out_pairs": len(pairs), "by_type": by_type,
              "noise_sigma": sigma, "per_plan_step": []}
    for it in range(plans):
        ns = capi.make_noise_spec(seed=0, iteration=it + 1, mode=capi.NOISE_SAMPLING, std0=sigma)
        nodes = pyoracle.noise_candidates(pm, ns, P, nominal, np.arange(0, N))
        nodes[0] = nominal  # candidate 0: the nominal itself (sampling/planner.cc)
        out = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, capi.SPLINE_ZERO, times, nodes, num_threads=threads)
        ng = m.scalars["ngeom"]
        xpos, xmat = np.zeros((N, H, ng, 3)), np.zeros((N, H, ng, 3, 3))
        for c in range(N):
            for t in range(H):
                s = out["states"][c, t]
                ph.set_state(s[:m.nq], s[m.nq:m.nq + m.nv], 0.0, mocap)
                ph.forward()
                xpos[c, t] = ph.get("geom_xpos", 3 * ng).reshape(ng, 3)
                xmat[c, t] = ph.get("geom_xmat", 9 * ng).reshape(ng, 3, 3)
        first = np.full(N, H)                      # first step at which some left-out pair is inside its margin
        closest = np.full(N, np.inf)
        hits = {}
        for p in pairs:
            d = pair_distance(m, p[0], p[1], xpos, xmat)
            closest = np.minimum(closest, d.min(axis=1) - margin[p])
            near = d < margin[p]
            if near.any():
                hits[p] = int(near.any(axis=1).sum())
                f = np.where(near.any(axis=1), near.argmax(axis=1), H)
                first = np.minimum(first, f)
        affected = first < H
        best = int(np.argmin(out["total_return"]))
        report["per_plan_step"].append({
            "plan_step": it, "affected_candidates": int(affected.sum()), "affected_frac": float(affected.mean()),
            "candidate_steps_after_first_missing_contact_frac": float(np.maximum(H - first, 0).sum() / (N * H)),
            "winner_affected": bool(affected[best]), "winner": best,
            "median_first_step_of_the_affected": (float(np.median(first[affected])) if affected.any() else None),
            "closest_approach_minus_margin_m": {"min": float(closest.min()), "p01": float(np.quantile(closest, 0.01)), "median": float(np.median(closest))},
            "pairs_hit": {f"{a}({TYPE_NAME[int(a_['geom_type'][a])]},body {int(a_['geom_bodyid'][a])})-{b}({TYPE_NAME[int(a_['geom_type'][b])]},body {int(a_['geom_bodyid'][b])})": n
                          for (a, b), n in sorted(hits.items(), key=lambda kv: -kv[1])[:12]},
        })
        nominal = nodes[best].copy()
        if advance:
            state = np.asarray(out["states"][best, 1], float)[:m.nq + m.nv]   # the plant follows the winner for one agent step
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
