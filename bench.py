#!/usr/bin/env python3
"""bench.py -- candidate-trajectory rollouts/sec of the MI355X rollout-and-evaluate path.

A "step" is one pass of the hot path over one batch of synthetic input: one Predictive
Sampling plan iteration's timed span in the reference (`rollouts_compute_time`,
mjpc/planners/sampling/planner.cc:169-191) = candidate noise + N rollouts of H steps +
selection of the best candidate, followed by the policy update that feeds the next step.

Workload at --gpus 1 = BASELINE.json configs[1]: Cartpole, Predictive Sampling,
4096 candidates, horizon 128, fp64, 10 cubic spline points. With --gpus N every rank
rolls out its own 4096 candidates of a global batch of N*4096 (weak scaling) and the
ranks exchange (best cost, index) + the winner's spline over RCCL.

Prints ONE JSON line (rank 0) with the fields of the driver contract plus `roofline`
and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--task", default="Cartpole")
    ap.add_argument("--candidates", type=int, default=4096, help="candidates per GPU")
    ap.add_argument("--horizon", type=int, default=128)
    ap.add_argument("--precision", type=int, default=64, choices=[32, 64])
    ap.add_argument("--planner", default="sampling", choices=["sampling", "cross_entropy"],
                    help="host planner driving the hot path (cross_entropy: BASELINE configs[2] with --task QuadrupedFlat)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def cpu_baseline(task, state, horizon, num_nodes, seconds, n_per_call, mocap=None, interp=None):
    """Times the CPU oracle (a port, NOT MuJoCo) driven through the reference's fan-out structure
    (one task per candidate, one physics arena per worker thread; sampling/planner.cc:355-393) on the
    GPU box's host cores. The thread count is the best of a short probe over {all, 1/2, 1/4, 1/8} of the
    visible cores (containers often expose more CPUs than their quota lets them run)."""
    from mujoco_mpc_amd import capi
    from oracle import pyoracle
    pm, pt = task.packed_model(), task.packed()
    cores = os.cpu_count() or 1
    dt = task.model.get_number("agent_timestep", task.model.timestep)
    times = np.array([k * (horizon - 1) * dt / (num_nodes - 1) for k in range(num_nodes)])
    rng = np.random.default_rng(0)
    nodes = np.clip(rng.normal(0, task.model.get_number("sampling_exploration", 0.5), (n_per_call, num_nodes, task.model.nu)), -1, 1)

    def run(n, threads):
        t0 = time.perf_counter()
        pyoracle.rollout_batch_fast(pm, pt, state, 0.0, mocap, n, horizon, num_nodes, capi.SPLINE_CUBIC if interp is None else interp,
                                    times, nodes[:n], num_threads=threads)
        return n / (time.perf_counter() - t0)

    run(64, 1)
    single = max(run(256, 1), run(256, 1))
    best_threads, best_rate = 1, 0.0
    for threads in sorted({max(1, cores // d) for d in (1, 2, 4, 8, 16)}):
        n = min(n_per_call, 64 * threads)
        run(n, threads)
        rate = max(run(n, threads), run(n, threads))
        if rate > best_rate:
            best_threads, best_rate = threads, rate
    done, t0 = 0, time.perf_counter()
    while True:
        run(n_per_call, best_threads)
        done += n_per_call
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return dict(value=done / el, unit="rollouts/s", cores=best_threads, kind="port",
                sample=f"{done} rollouts of H={horizon} ({el:.1f} s) through the C oracle's ThreadPool-style fan-out, "
                       f"{best_threads} threads (best of a probe over {cores} visible CPUs; 1 thread: {single:.0f} rollouts/s); "
                       f"gcc -O3 -march=native -flto; CPU restatement, not MuJoCo")


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    import torch
    from mujoco_mpc_amd import capi
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.task import load_task

    group = None
    if world > 1:
        import torch.distributed as dist
        from mujoco_mpc_amd.distributed import RankGroup
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        group = RankGroup(dist, torch.device("cuda", local_rank))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    task = load_task(args.task)
    model = task.model
    # the C++ mjpc::GpuSamplingPlanner (mujoco_mpc_amd/host) drives the C ABI; Python only lends it
    # torch.distributed as the transport of the per-step candidate exchange when there are several ranks
    H = args.horizon
    planner = HostPlanner(task, device=local_rank, precision=args.precision, seed=0,
                          num_trajectory=args.candidates * world,  # lifts kMaxTrajectory = 128
                          group=group, kind=args.planner)
    # synthetic initial condition: the task's home keyframe
    home = model.keyframes.get("home")
    qpos = np.array(home["qpos"] if home else model.qpos0, float)
    qvel = np.array(home["qvel"] if home else np.zeros(model.nv), float)
    mocap_pos = mocap_quat = None
    if model.nmocap:  # mocap bodies at their model pose (State::Reset)
        ids = [b for b in range(model.nbody) if model.arrays["body_mocapid"][b] >= 0]
        ids.sort(key=lambda b: model.arrays["body_mocapid"][b])
        mocap_pos = np.array([model.arrays["body_pos"][b] for b in ids], float)
        mocap_quat = np.array([model.arrays["body_quat"][b] for b in ids], float)
    if args.task == "HumanoidTrack":
        # Task::Transition edits the simulation state: first keyframe of the motion, interpolated marker positions
        mode = 9   # Walk
        planner.task_transition_state(0.0, mode, qpos, qvel, mocap_pos.reshape(-1))
        task.transition(0.0, mode)   # the Python mirror keeps the frozen residual state for the cpu_baseline leg
    elif hasattr(task, "transition"):
        planner.task_transition(0.0)
    planner.reset(H)
    P = planner.num_spline_points
    planner.set_state(qpos, qvel, 0.0, mocap_pos=mocap_pos, mocap_quat=mocap_quat)

    def step():
        planner.optimize_policy(H)

    def fence():
        if group is not None:
            group.barrier()
        planner.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    planner.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = planner.timing_read()
    if group is not None:
        elapsed = group.max_scalar(elapsed)

    total_rollouts = args.candidates * world * args.steps
    value = total_rollouts / elapsed
    if rank == 0:
        bytes_per_rollout = planner.algorithmic_bytes(H, P)
        bytes_per_launch = bytes_per_rollout * args.candidates
        avg_kernel_s = kernel_ms / max(launches, 1) * 1e-3
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        out = {
            "metric": "candidate-trajectory rollouts/sec (fixed horizon)",
            "value": value, "unit": "rollouts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.precision == 64 else "f32", "data": "synthetic",
            "config": {"workload": (f"{args.task} Predictive Sampling, {args.candidates} candidates/GPU, horizon {H}, "
                                    f"{P} cubic spline points, fp{args.precision} " +
                                    {"Cartpole": "(BASELINE.json configs[1])", "HumanoidTrack": "(BASELINE.json configs[3]: one GPU's 8192-candidate share, fp64 instead of fp32)"}.get(args.task, "")) if args.planner == "sampling" else
                                   (f"{args.task} Cross-Entropy, {args.candidates} candidates/GPU, horizon {H}, {P} zero-order spline "
                                    f"points, fp{args.precision} (BASELINE.json configs[2] at --task QuadrupedFlat --candidates 16384 --horizon 100)"),
                       "candidates_per_gpu": args.candidates, "horizon": H, "spline_points": P,
                       "parallelism": f"candidates sharded over {world} rank(s)", "kernel": planner.kernel_name, "host": ("C++ mjpc::GpuSamplingPlanner" if args.planner == "sampling" else "C++ mjpc::GpuCrossEntropyPlanner") + " over the C ABI"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": avg_kernel_s * 1e3, "bytes_per_launch": bytes_per_launch,
                         "note": "algorithmic bytes (SURVEY 8d) / HIP-event time of one rollout on the context's stream "
                                 "(lane-per-candidate models: its three launches -- time loop, sensor stage, returns; "
                                 "profiles/ lists each); issue/latency-bound at this batch size, see DESIGN.md"},
        }
        # HBM traffic per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # collected in separate --pmc runs as MI355X_MICROARCH.md prescribes); only for the profiled workload
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            if args.task == "Cartpole" and args.candidates == 4096 and H == 128 and args.precision == 64:
                out["roofline"]["traffic"] = pmc["n4096"]["hbm_bytes_per_launch"]
            if args.task == "QuadrupedFlat" and args.candidates == 16384 and H == 100:
                pq = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_quadruped.json")))
                out["roofline"]["traffic"] = pq[f"fp{args.precision}"]["derived"]["hbm_bytes_per_launch"]
            if args.task == "HumanoidTrack" and args.candidates == 8192 and H == 64:
                ph = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_humanoid.json")))
                out["roofline"]["traffic"] = ph[f"fp{args.precision}"]["derived"]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        if not args.no_cpu_baseline:
            st = np.concatenate([qpos, qvel])
            # same workload (model, horizon, spline), batch enlarged so that every host thread has
            # >= 64 rollouts per fan-out and thread start-up does not dominate the CPU number
            n_cpu = max(args.candidates, 64 * (os.cpu_count() or 1))
            out["cpu_baseline"] = cpu_baseline(task, st, H, P, args.cpu_seconds, n_cpu,
                                               mocap=None if mocap_pos is None else np.hstack([mocap_pos, mocap_quat]).reshape(-1),
                                               interp=capi.SPLINE_CUBIC if args.planner == "sampling" else capi.SPLINE_ZERO)
        print(json.dumps(out), flush=True)
    if group is not None:
        group.barrier()
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
