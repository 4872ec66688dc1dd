"""fp32 vs fp64 wave kernel vs the fp64 oracle: relative error of the total returns and the worst state error of one candidate
at horizons 5 / 40 / 100 on the Quadruped and the Humanoid (numbers quoted in DESIGN.md 4.5)."""
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

