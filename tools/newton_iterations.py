#!/usr/bin/env python3
"""Newton iterations per step of the north-star workload, replayed on the CPU oracle (same solver, same stopping rule as the device):
24 candidates x 99 steps from the home keyframe with the task's exploration noise -- DESIGN.md 4.6 quotes the mean."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle
t=load_task("QuadrupedFlat"); t.transition(0.0)
pm,pt=t.packed_model(),t.packed()
m=pm.struct
home=t.model.keyframes["home"]["qpos"]
mocap=np.array([0.3,0,0.26,1,0,0,0,-2.5,0,0,1,0,0,0.0])
std=t.model.get_number("sampling_exploration",0.1)
print("std",std)
rng=np.random.default_rng(0)
H,P=100,3
its=[]; ncon=[]
for c in range(24):
    nodes=np.clip(rng.normal(0, 0.5*2*std if False else std, (P,12)),-1,1) if c>0 else np.zeros((P,12))
    ph=pyoracle.Physics(pm)
    ph.set_state(home,np.zeros(18),0.0,mocap)
    times=np.arange(P)*(H-1)*0.01/(P-1)
    for s in range(H-1):
        k=np.searchsorted(times, s*0.01, side='right')-1
        ph.set_ctrl(nodes[max(k,0)])
        ph.step()
        its.append(int(ph.get("solver_iter")[0])); ncon.append(int(ph.get("ncon")[0]))
its=np.array(its)
print("mean iters",its.mean(),"median",np.median(its),"hist",np.bincount(its)[:20], "mean ncon", np.mean(ncon))
