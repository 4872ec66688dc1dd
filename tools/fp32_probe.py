import numpy as np, sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle
def mocap7(mpos): return np.concatenate([np.concatenate([p, [1, 0, 0, 0]]) for p in np.asarray(mpos).reshape(-1, 3)])
for name in ("QuadrupedFlat", "HumanoidTrack"):
    t = load_task(name)
    if name == "QuadrupedFlat":
        t.transition(0.0); q = t.model.keyframes["home"]["qpos"]; v = np.zeros(18); mocap = np.array([0.3,0,0.26,1,0,0,0,-2.5,0,0,1,0,0,0.]); std=0.3
    else:
        e = t.transition(0.0, mode=9); q, v, mocap = e["qpos"], e["qvel"], mocap7(e["mocap_pos"]); std=0.3
    pm, pt = t.packed_model(), t.packed()
    N, P = 16, 4
    rng = np.random.default_rng(0)
    for H in (5, 40, 100):
        dt = t.model.get_number("agent_timestep", t.model.timestep)
        times = np.arange(P) * max((H - 1) * dt / (P - 1), 1e-3)
        nodes = np.clip(rng.normal(0, std, (N, P, t.model.nu)), -1, 1)
        state = np.concatenate([q, v])
        ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 1, times, nodes, num_threads=8)
        out = {}
        for prec in (64, 32):
            ctx = capi.Context(pm, pt, 0, prec)
            ctx.set_state(state, 0.0, mocap)
            ctx.rollout_splines(H, 1, times, nodes)
            ret, fail = ctx.returns()
            tr = ctx.fetch_trajectory(3)
            out[prec] = (ret.copy(), fail.copy(), tr.states.copy())
            ctx.close()
        r64, r32 = out[64][0], out[32][0]
        print(name, "H", H, "fail32", out[32][1].sum(), "ret rel err 32 vs oracle: max %.2e med %.2e" % (np.max(np.abs(r32-ref["total_return"])/np.abs(ref["total_return"])), np.median(np.abs(r32-ref["total_return"])/np.abs(ref["total_return"]))),
              "| 64: %.1e" % np.max(np.abs(r64-ref["total_return"])/np.abs(ref["total_return"])), "| state err32 max %.2e" % np.max(np.abs(out[32][2]-ref["states"][3])))

# where does the Quadruped fp32 return drift at H = 100 come from?
t = load_task("QuadrupedFlat"); t.transition(0.0)
q = t.model.keyframes["home"]["qpos"]; v = np.zeros(18); mocap = np.array([0.3,0,0.26,1,0,0,0,-2.5,0,0,1,0,0,0.])
pm, pt = t.packed_model(), t.packed()
N, P, H = 4, 4, 100
rng = np.random.default_rng(0)
times = np.arange(P) * (H - 1) * 0.01 / (P - 1)
nodes = np.clip(rng.normal(0, 0.3, (N, P, 12)), -1, 1)
state = np.concatenate([q, v])
ref = pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, N, H, P, 1, times, nodes, num_threads=4)
ctx = capi.Context(pm, pt, 0, 32)
ctx.set_state(state, 0.0, mocap); ctx.rollout_splines(H, 1, times, nodes)
tr = ctx.fetch_trajectory(1)
dc = np.abs(tr.costs - ref["costs"][1])
print("cost err by step (every 10):", np.array2string(dc[::10], precision=2))
print("costs ref (every 10):", np.array2string(ref["costs"][1][::10], precision=4))
dr = np.abs(tr.residual - ref["residual"][1])
print("residual err max per entry at t=99:", np.array2string(dr[99], precision=1))
print("state err at t=99", np.abs(tr.states[99]-ref["states"][1][99]).max(), "times err", np.abs(tr.times-ref["times"][1]).max())
print("ref costs 84..99", np.array2string(ref["costs"][1][84:], precision=3))
print("f32 costs 84..99", np.array2string(tr.costs[84:], precision=3))
print("ref times 84..99", np.array2string(ref["times"][1][84:], precision=3))
print("f32 times 84..99", np.array2string(tr.times[84:], precision=3))
print("shapes", ref["costs"].shape, tr.costs.shape, tr.states.shape, ref["states"].shape)
ret32, fail32 = ctx.returns()
print("ref failure", ref["failure"], "ref returns", ref["total_return"], "f32 fail", fail32, "f32 returns", ret32)
c64 = capi.Context(pm, pt, 0, 64); c64.set_state(state, 0.0, mocap); c64.rollout_splines(H, 1, times, nodes)
print("f64 returns", c64.returns())
