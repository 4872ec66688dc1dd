import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.task import load_task
import test_gpu_quad as T
quad = load_task("QuadrupedFlat"); quad.transition(0.0)
N, H, P = 200, 60, 3
times = np.arange(P) * ((H - 1) * 0.01 / (P - 1))
nominal = np.clip(np.random.default_rng(4).normal(0, 0.05, (P, 12)), -1, 1)
ns = capi.make_noise_spec(seed=5, iteration=1, mode=capi.NOISE_SAMPLING, std0=0.1)
ref = None
for cpw in (16, 8, 4, 2, 1):
    ctx = T.context(quad, {"MJPCX_QUAD_CPW": str(cpw)})
    ctx.rollout_noise(N, H, 0, times, nominal, ns)
    ret, fail = ctx.returns()
    ctx.close()
    if ref is None: ref = ret.copy()
    d = np.abs(ret - ref) / (1 + np.abs(ref))
    print(cpw, "max rel diff", d.max(), "n differing", int((d > 0).sum()), "which", np.nonzero(d > 0)[0][:10])
