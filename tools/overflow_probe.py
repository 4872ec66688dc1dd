#!/usr/bin/env python3
"""How many rollouts of a full-size launch overflow the first pass's contact lists (MJPCX_TREE_ONE_PASS=1) and how many
fail after the second pass: python tools/overflow_probe.py [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
quad = load_task("QuadrupedFlat"); quad.transition(0.0)
pm, pt = quad.packed_model(), quad.packed()
state = np.concatenate([quad.model.keyframes["home"]["qpos"], np.zeros(18)])
mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0])
H, P = 100, 3
times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
nominal = np.clip(np.random.default_rng(5).normal(0, 0.2, (P, 12)), -1, 1)
var = np.full(P * 12, 0.1 ** 2)
ns = capi.make_noise_spec(seed=11, iteration=3, mode=capi.NOISE_CROSS_ENTROPY, std0=0.1, param_variance=var, explore_count=N // 10, std1=0.01)
ctx = capi.Context(pm, pt, 0, 64)
ctx.set_state(state, 0.0, mocap)
ctx.timing_reset()
ctx.rollout_noise(N, H, 0, times, nominal, ns)
ret, fail = ctx.returns()
ms, n = ctx.timing_read()
raw = ctx.failure_raw
print(ctx.kernel_name[:40], "failures", int(fail.sum()), "of", N, "bits", sorted(set(hex(int(x) >> 8 & 0xff) for x in raw[raw != 0])), "kernel ms", ms / max(n, 1))
