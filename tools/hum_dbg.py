"""Bring-up aid: the Humanoid walk of tests/test_gpu_humanoid.py on the Jacobian-free path and (MJPCX_NO_TREE=1) on the row-table path, per
candidate: raw failure words (warning bits << 8, step << 16), return error and the first step whose state leaves 1e-6 of the oracle."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle
def mocap7(mpos): return np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(mpos).reshape(-1, 3)])
t = load_task("HumanoidTrack"); e = t.transition(0.0, mode=9)
pm, pt = t.packed_model(), t.packed()
state = np.concatenate([e["qpos"], e["qvel"]]); mocap = mocap7(e["mocap_pos"])
N, H, P, interp, seed, std = 6, 40, 8, 0, 3, 0.3
rng = np.random.default_rng(seed)
dt = t.model.get_number("agent_timestep", t.model.timestep)
times = np.arange(P) * max((H - 1) * dt / max(P - 1, 1), 1e-3)
nodes = np.clip(rng.normal(0, std, (N, P, t.model.nu)), -1, 1)
ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, interp, times, nodes, num_threads=8)
for env in ({}, {"MJPCX_NO_TREE": "1"}):
    os.environ.pop("MJPCX_NO_TREE", None); os.environ.update(env)
    ctx = capi.Context(pm, pt, 0, 64)
    ctx.set_state(state, 0.0, mocap)
    ctx.rollout_splines(H, interp, times, nodes)
    ret, fail = ctx.returns()
    print("raw failure", [hex(int(x)) for x in ctx.failure_raw]); print(env, ctx.kernel_name[:60], "fail", fail, "ret err", np.abs(ret - ref["total_return"]) / (1 + np.abs(ref["total_return"])))
    for c in range(N):
        tr = ctx.fetch_trajectory(c)
        d = np.abs(tr.states - ref["states"][c]).max(axis=1)
        bad = np.nonzero(d > 1e-6)[0]
        print("  cand", c, "first step with state err > 1e-6:", bad[:1], "max", d.max())
    ctx.close()
