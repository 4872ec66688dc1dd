"""-m gpu: the C++ planner (driven through hostplanner.py) against the Python mirror, and its sharded mode:
two processes sharing ONE GPU (gloo as the exchange transport) must reproduce the single-rank plan exactly."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_planner_equals_python_planner(cartpole):
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.planners import GpuSamplingPlanner, State
    H, N = 40, 512
    cpp = HostPlanner(cartpole, seed=3, num_trajectory=N)
    cpp.reset(H)
    py = GpuSamplingPlanner(seed=3)
    py.initialize(cartpole.model, cartpole); py.num_trajectory_ = N; py.allocate(); py.reset(H)
    st = State(cartpole.model)
    for k in range(5):
        q, v, t = [0.1 * k, 0.5], [0.0, -0.1], 0.04 * k
        st.set(q, v, time=t); py.set_state(st); py.optimize_policy(H)
        cpp.set_state(q, v, t); cpp.optimize_policy(H)
        assert cpp.winner == py.winner
        assert cpp.best_score == py.candidate_score(0) and cpp.improvement == py.improvement
        ct, cv = cpp.policy()
        assert np.array_equal(ct, py.policy.plan.times()) and np.array_equal(cv, py.policy.plan.values())
        a = np.zeros(1)
        py.action_from_policy(a, None, t + 0.013)
        assert np.array_equal(cpp.action(t + 0.013), a)
    assert "StaticCartpole" in cpp.kernel_name


WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from mujoco_mpc_amd.distributed import RankGroup
from mujoco_mpc_amd.hostplanner import HostPlanner
from mujoco_mpc_amd.task import load_task
world = int(os.environ.get("WORLD_SIZE", "1"))
group = None
if world > 1:
    dist.init_process_group(backend="gloo")
    group = RankGroup(dist, torch.device("cpu"))
task = load_task("Cartpole")
p = HostPlanner(task, device=0, seed=7, num_trajectory=1000, group=group)
H = 32
p.reset(H)
log = []
for k in range(4):
    p.set_state([0.05 * k, 0.3], [0.0, 0.1], 0.04 * k)
    p.optimize_policy(H)
    t, v = p.policy()
    log.append(dict(winner=p.winner, score=p.best_score, improvement=p.improvement, plan=v.tolist()))
if group is None or group.rank == 0:
    print("RESULT " + json.dumps(log))
if group is not None:
    dist.barrier(); dist.destroy_process_group()
'''


def run(world):
    script = WORKER % dict(root=ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-c", script] if world == 1 else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
        "127.0.0.1", "--master-port", "29544", "--no-python", sys.executable, "-c", script]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_two_ranks_on_one_gpu_equal_one_rank():
    one, two = run(1), run(2)
    assert one == two
