#!/usr/bin/env python3
"""Which candidates of the bench workload does rollout_quad_kernel hand on, when and why? Runs the planner for a few iterations (as bench.py's
warm-up does), then rolls the batch out around its nominal with MJPCX_QUAD_NO_FALLBACK=1 and reads reason and step from failure[]; the
oracle names the geoms of the first moving-geom contact of a few handed-on candidates."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.hostplanner import HostPlanner
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
N, H = 16384, 100
task = load_task("QuadrupedFlat")
planner = HostPlanner(task, device=0, precision=64, seed=0, num_trajectory=N, kind="sampling")
qpos, qvel, mocap_pos, mocap_quat = bench.initial_condition("QuadrupedFlat", task, planner)
planner.reset(H)
P = planner.num_spline_points
planner.set_state(qpos, qvel, 0.0, mocap_pos=mocap_pos, mocap_quat=mocap_quat)
for _ in range(iters):
    planner.optimize_policy(H)
times, nominal = planner.policy()
print("nominal nodes after", iters, "iterations: |max|", np.abs(nominal).max(), "handed on in the last step", planner.quad_stats())
planner.close()
os.environ["MJPCX_QUAD_NO_FALLBACK"] = "1"
t2 = load_task("QuadrupedFlat"); t2.transition(0.0)
pm, pt = t2.packed_model(), t2.packed()
ctx = capi.Context(pm, pt, 0, 64)
state = np.concatenate([qpos, qvel]); mocap = np.hstack([mocap_pos, mocap_quat]).reshape(-1)
ctx.set_state(state, 0.0, mocap)
std = task.model.get_number("sampling_exploration", 0.1)
ns = capi.make_noise_spec(seed=1, iteration=iters + 1, mode=capi.NOISE_SAMPLING, std0=std)
ctx.rollout_noise(N, H, 0, times, nominal, ns)
ctx.returns()
raw = ctx.failure_raw
fl = raw[(raw & 0x40000000) != 0]
print("flagged", len(fl), "of", N)
reasons = collections.Counter(int(x) & 0x3f for x in fl)
print("reason bits (1 overflow, 2 leg-leg, 4 notPD, 8 bad, 16 limits, 32 trunk-leg):", dict(reasons))
steps = np.array([(int(x) >> 8) & 0x1ff for x in fl])
print("flag step: mean %.1f median %.0f  histogram by decile:" % (steps.mean(), np.median(steps)), np.histogram(steps, bins=10, range=(0, 100))[0])
print("mean remaining steps", (H - steps).mean())
# which pairs: oracle replay of a few flagged candidates up to their flag step
idx = np.nonzero((raw & 0x40000000) != 0)[0][:12]
nodes = pyoracle.noise_candidates(pm, ns, P, nominal, idx)
m = pm.struct
gb = [m.geom_bodyid[g] for g in range(m.ngeom)]
pairs = collections.Counter()
for k, c in enumerate(idx):
    ph = pyoracle.Physics(pm); ph.set_state(qpos, qvel, 0.0, mocap)
    st = (int(raw[c]) >> 8) & 0x1ff
    for s in range(st + 1):
        seg = np.searchsorted(times, s * 0.01, side="right") - 1
        ph.set_ctrl(nodes[k][max(seg, 0)])
        if s == st:
            ph.forward()
            con = ph.get("contact").reshape(-1, 11)
            for r in con:
                if int(r[7]) > 3:
                    pairs[(int(r[7]), int(r[8]), gb[int(r[7])], gb[int(r[8])], int(r[9]))] += 1
        else:
            ph.step()
print("moving-geom contacts at the flag step (geom1, geom2, body1, body2, dim):", dict(pairs))
