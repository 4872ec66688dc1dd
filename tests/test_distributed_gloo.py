"""world_size-2 `gloo` test of the candidate sharding + exchange (mujoco_mpc_amd/distributed.py):
two ranks, each rolling out half of the candidates, must produce the same policy as one rank rolling
out all of them (the noise is counter-based on the GLOBAL candidate index)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
from mujoco_mpc_amd.distributed import RankGroup
from mujoco_mpc_amd.planners import GpuSamplingPlanner, State
from mujoco_mpc_amd.task import load_task
from oracle_backend import OracleContext
world = int(os.environ.get("WORLD_SIZE", "1"))
group = None
if world > 1:
    dist.init_process_group(backend="gloo")
    group = RankGroup(dist, torch.device("cpu"))
task = load_task("Cartpole")
p = GpuSamplingPlanner(seed=5, group=group, backend_factory=lambda t: OracleContext(t, threads=1))
p.initialize(task.model, task); p.num_trajectory_ = 48; p.allocate()
H = 24
p.reset(H)
st = State(task.model); st.set([0.2, 0.6], [0.0, 0.1])
p.set_state(st)
log = []
for it in range(3):
    p.optimize_policy(H)
    log.append(dict(winner=int(p.winner), score=p.candidate_score(0), improvement=p.improvement,
                    plan=p.policy.plan.values().tolist()))
# Cross-Entropy: sharded candidates + all-gathered elites + all-reduced moments
from mujoco_mpc_amd.planners import GpuCrossEntropyPlanner
ce = GpuCrossEntropyPlanner(seed=8, group=group, backend_factory=lambda t: OracleContext(t, threads=1))
ce.initialize(task.model, task); ce.num_trajectory_ = 40; ce.n_elite_ = 6; ce.allocate()
ce.std_initial_, ce.std_min_, ce.explore_fraction_ = 0.3, 0.05, 0.2
ce.reset(H); ce.set_state(st)
for it in range(2):
    ce.optimize_policy(H)
    log.append(dict(winner=ce.trajectory_order[0], score=float(ce.improvement), improvement=float(ce.variance[:10].sum()),
                    plan=ce.policy.plan.values().tolist()))
# iLQG derivative sweep sharded over the time steps (RankGroup.sharded_transition_fd); T = 7 is not divisible by 2
from oracle import pyoracle
pm, pt = task.packed_model(), task.packed()
rng = np.random.default_rng(3)
T = 7
fd_times, fd_states, fd_actions = np.arange(T) * 0.01, rng.normal(0, 0.3, (T, 4)), rng.uniform(-1, 1, (T, 1))
fd = lambda tt, ss, aa, **kw: pyoracle.transition_fd(pm, pt, ss, tt, aa, **kw)
full = fd(fd_times, fd_states, fd_actions, centered=1)
if group is not None:
    assert group.time_shard(7) == ((0, 4) if group.rank == 0 else (4, 7))
    sharded = group.sharded_transition_fd(fd, fd_times, fd_states, fd_actions, centered=1)
    for a, b in zip(full, sharded):
        assert a.shape == b.shape and np.array_equal(a, b)
    one_step = group.sharded_transition_fd(fd, fd_times[:1], fd_states[:1], fd_actions[:1], centered=1)  # rank 1's share is empty
    assert all(np.array_equal(a[:1], b) for a, b in zip(full, one_step))
log.append(dict(winner=0, score=float(full[0].sum()), improvement=float(full[2].sum()), plan=full[1].reshape(-1).tolist()))
if group is None or group.rank == 0:
    print("RESULT " + json.dumps(log))
if group is not None:
    # unit checks of the exchange primitives
    idx, ret = group.merge_topk(np.array([group.rank * 10 + 1, group.rank * 10 + 2]), np.array([1.0 - group.rank, 5.0]), 3)
    assert list(idx) == [11, 1, 2] and list(ret) == [0.0, 1.0, 5.0], (idx, ret)
    assert group.broadcast_scalar(3.5 if group.rank == 1 else None, src=1) == 3.5
    assert group.owner_of(47, 48) == 1 and group.owner_of(0, 48) == 0
    assert group.max_scalar(float(group.rank)) == 1.0
    dist.barrier(); dist.destroy_process_group()
'''


def run(world):
    script = WORKER % dict(root=ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    if world == 1:
        cmd = [sys.executable, "-c", script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29533", "--no-python", sys.executable, "-c", script]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    import json
    return json.loads(line[7:])


def test_two_ranks_equal_one_rank():
    one, two = run(1), run(2)
    assert len(one) == len(two) == 6
    for k, (a, b) in enumerate(zip(one, two)):
        assert a["winner"] == b["winner"]
        if k < 3 or k == 5:   # Predictive Sampling, and the sharded iLQG derivative sweep: bit-identical
            assert a["score"] == b["score"] and a["improvement"] == b["improvement"]
            assert np.array_equal(np.array(a["plan"]), np.array(b["plan"]))
        else:       # Cross-Entropy: sums are re-associated across ranks
            assert abs(a["score"] - b["score"]) < 1e-12 and abs(a["improvement"] - b["improvement"]) < 1e-12
            assert np.allclose(np.array(a["plan"]), np.array(b["plan"]), rtol=0, atol=1e-14)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_launcher_dry_run_at_world_size_two(scaling):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank, 127.0.0.1), with
    --dry-run taking the device out: rendezvous, the communicator-id broadcast, the candidate split and the max-over-ranks timing
    run for real; rank 0 prints exactly one JSON line."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run",
           "--candidates", "4097", "--scaling", scaling]  # (strong: 4097 is the global batch, 2049 + 2048; weak: each rank's)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["dry_run"] is True and line["scaling"] == scaling
    assert line["ms_per_step"] >= 20.0          # the slower rank (2 x 10 ms per step), not rank 0's own 10 ms


def test_bench_launches_its_own_ranks_when_called_plainly():
    """`python bench.py --gpus 2 ...` with NO launcher around it (how the driver's BENCH record invokes it): bench.py starts one copy
    of itself per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would), rank 0's JSON line comes out
    once, the exit code is 0 -- and non-zero when a rank dies."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--candidates", "4097"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["dry_run"] is True
    assert line["ms_per_step"] >= 20.0
    # a rank that dies takes the launch down with a non-zero exit code instead of leaving the others at a barrier
    env["MJPC_BENCH_DRY_RUN_FAIL_RANK"] = "1"
    env["MJPC_BENCH_LAUNCH_TIMEOUT_S"] = "120"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


# ---------------------------------------------------------------- the C++ planners (what bench.py ships), on a CPU stand-in for the device
CPP_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from mujoco_mpc_amd.distributed import RankGroup
from mujoco_mpc_amd.hostplanner import HostPlanner
from mujoco_mpc_amd.task import load_task
world = int(os.environ.get("WORLD_SIZE", "1"))
group = None
if world > 1:
    dist.init_process_group(backend="gloo")
    group = RankGroup(dist, torch.device("cpu"))
task = load_task("Particle")
log = []
for kind, n in (("sampling", 49), ("cross_entropy", 40), ("robust", 45)):      # 49, 45: the first rank holds one candidate more
    p = HostPlanner(task, seed=5, num_trajectory=n, kind=kind, group=group)
    import ctypes as C
    glob = C.CDLL(None)                                         # the process's global symbol scope: the preloaded stand-in comes first
    glob.mjpcx_kernel_name.restype = C.c_char_p
    glob.mjpcx_kernel_name.argtypes = [C.c_void_p]
    assert b"cpu stub" in glob.mjpcx_kernel_name(p._ctx())      # ... and it is what the C++ planner's context is made of
    if kind == "cross_entropy":
        p.ce_set(n_elite=6, std_initial=0.3, std_min=0.05, explore_fraction=0.2)
    if kind == "robust":                                        # 7 ranked candidates x 3 perturbed rollouts = 21: 11 + 10
        p.robust_config(ncandidates=7, nrepetitions=3, xfrc_std=0.4, xfrc_rate=0.1)
    H = 20
    p.reset(H)
    p.set_state(np.array([0.1, -0.05]), np.array([0.0, 0.02]), 0.0, mocap_pos=np.array([[0.2, 0.1, 0.01]]), mocap_quat=np.array([[1.0, 0, 0, 0]]))
    for it in range(3):
        p.optimize_policy(H)
        times, values = p.policy()
        entry = dict(kind=kind, winner=int(p.winner), score=float(p.best_score), plan=np.asarray(values).reshape(-1).tolist())
        if kind == "robust":
            best, scores = p.robust_result(7)
            entry.update(best_candidate=int(best), perturbed=np.asarray(scores).tolist())
        log.append(entry)
    p.close()
if group is None or group.rank == 0:
    print("RESULT " + json.dumps(log))
if group is not None:
    dist.barrier(); dist.destroy_process_group()
'''


def run_cpp(world):
    from mujoco_mpc_amd.build import build_host
    build_host()
    stub_dir = os.path.join(ROOT, "tests", "stub")
    so = os.path.join(stub_dir, "libmjpcx_stub.so")
    src = os.path.join(stub_dir, "mjpcx_stub.c")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-Wall", "-o", so, src, "-lm"])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", LD_PRELOAD=so)
    script = CPP_WORKER % dict(root=ROOT)
    if world == 1:
        cmd = [sys.executable, "-c", script]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", "29541", "--no-python", sys.executable, "-c", script]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    import json
    return json.loads(line[7:])


def test_cpp_planners_two_ranks_equal_one_rank_on_a_cpu_stand_in():
    """The C++ planners bench.py drives (host/mjpc/planners/gpu_sampling, gpu_cross_entropy) at world size 1 and 2 over gloo, WITHOUT a
    device: tests/stub/mjpcx_stub.c is preloaded in place of libmjpcx.so -- a stand-in whose "rollout" is a fixed function of the
    candidate's spline and whose noise is keyed on the global candidate index, i.e. the sharding contract and nothing else. What is under
    test is the planners' own sharding: contiguous ranges (49 candidates: 25 + 24), candidate_offset, the exchange / merge / sum callbacks.
    The winner, its score and the policy must not depend on the number of ranks."""
    one, two = run_cpp(1), run_cpp(2)
    assert len(one) == len(two) == 9
    for a, b in zip(one, two):
        assert a["kind"] == b["kind"] and a["winner"] == b["winner"]
        if a["kind"] == "sampling":   # bit-identical
            assert a["score"] == b["score"] and a["plan"] == b["plan"]
        elif a["kind"] == "robust":   # the ranked candidates of all ranks merged, the perturbed rollouts split 11 + 10: bit-identical
            assert a["best_candidate"] == b["best_candidate"] and a["perturbed"] == b["perturbed"] and a["plan"] == b["plan"]
            assert len(a["perturbed"]) == 7 and len(set(a["perturbed"])) == 7
        else:                         # elite sums are re-associated across ranks
            assert abs(a["score"] - b["score"]) < 1e-12 and np.allclose(a["plan"], b["plan"], rtol=0, atol=1e-13)
