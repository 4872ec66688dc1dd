import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from mujoco_mpc_amd.task import load_task
from oracle import pyoracle
t = load_task("QuadrupedFlat"); t.transition(0.0)
pm, pt = t.packed_model(), t.packed()
home = t.model.keyframes["home"]["qpos"]; state = np.concatenate([home, np.zeros(18)])
MOCAP = np.array([0.3,0,0.26,1,0,0,0,-2.5,0,0,1,0,0,0])
H, P = 100, 3
times = np.arange(P) * (H - 1) * 0.01 / (P - 1)
rng = np.random.default_rng(0)
print("cpus", os.cpu_count())
for threads in (1, 32, 64, 128, 256):
    N = max(8 * threads, 16)
    nodes = np.clip(rng.normal(0, 0.04, (N, P, 12)), -1, 1)
    t0 = time.perf_counter()
    pyoracle.rollout_batch_fast(pm, pt, state, 0.0, MOCAP, N, H, P, 0, times, nodes, num_threads=threads)
    el = time.perf_counter() - t0
    print(f"cpu port: {threads} threads, {N} rollouts x {H} steps in {el:.2f} s = {N*H/el/1e6:.3f} M steps/s, {N/el:.0f} rollouts/s", flush=True)
