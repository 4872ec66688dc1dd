#!/usr/bin/env python3
"""bench.py -- candidate-trajectory rollouts/sec of the MI355X rollout-and-evaluate path.

A "step" is one pass of the hot path over one batch of synthetic input: one Predictive
Sampling plan iteration's timed span in the reference (`rollouts_compute_time`,
mjpc/planners/sampling/planner.cc:169-191) = candidate noise + N rollouts of H steps +
selection of the best candidate, followed by the policy update that feeds the next step.

Headline workload (--gpus 1, no other flags) = the workload BASELINE.json's `north_star` states its target
on: **Quadruped (Unitree A1, flat terrain) Predictive Sampling, 16384 candidates, horizon 100, fp64**
(model and batch of BASELINE configs[2]; the reference's arithmetic type). With --gpus N every rank rolls
out its own 16384 candidates of a global batch of N*16384 (weak scaling) and the ranks exchange
(best cost, index) + the winner's spline.

The same JSON line carries, in `extra`, one entry per further single-GPU BASELINE config so that one
driver run covers them all: configs[1] (Cartpole PS 4096 x 128 fp64), configs[2] through the Cross-Entropy
planner, configs[3] (Humanoid tracking PS, one GPU's 8192-candidate share, fp32) and configs[4] (one iLQG
iteration on the Quadruped, with the CPU port's iteration time beside it).

Prints ONE JSON line (rank 0) with the fields of the driver contract plus `roofline`, `cpu_baseline`, `extra`.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6   # MI355X vector FP64 (not in the local guide)
FP32_VALU_PEAK_TF = 157.3  # MI355X vector FP32 (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--task", default="QuadrupedFlat")
    ap.add_argument("--candidates", type=int, default=0, help="candidates per GPU (0: the BASELINE size of the task)")
    ap.add_argument("--horizon", type=int, default=0, help="0: the BASELINE horizon of the task")
    ap.add_argument("--precision", type=int, default=64, choices=[32, 64])
    ap.add_argument("--planner", default="sampling", choices=["sampling", "cross_entropy"],
                    help="host planner driving the hot path")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank rolls out --candidates; strong: --candidates is the GLOBAL batch (BASELINE configs[2]: "
                         "16384 candidates, 1 -> 8 GPUs), split over the ranks in contiguous ranges")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="one GPU: time the share of every rank of a W-rank strong-scaling run in turn (N / W candidates at that rank's "
                         "candidate offset) -- what the per-rank time of --scaling strong will be before the hardware shows up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` configs (they only run at --gpus 1)")
    ap.add_argument("--no-box-probe", action="store_true", help="skip gpu_clock.box_probe (three library-kernel figures that tell the pool's kinds of box apart)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test without a device: rendezvous over gloo, the unique-id broadcast, the candidate split, the "
                         "barrier + max-over-ranks timing and the JSON line -- everything of the --gpus N path except the kernels")
    return ap.parse_args()


# BASELINE.json sizes per task: (candidates per GPU, horizon, label)
BASELINE_SIZE = {
    "Cartpole": (4096, 128, "BASELINE.json configs[1]"),
    "QuadrupedFlat": (16384, 100, "north_star workload; model and batch of BASELINE.json configs[2]"),
    "HumanoidTrack": (8192, 64, "BASELINE.json configs[3]: one GPU's 8192-candidate share of 65536"),
}


def kernel_source_sha16(unit=None):
    """identity of the device code being benchmarked: sha256 over the kernel sources (csrc/*.h, *.hip, generated/*, include/mjpcx.h) and build.py (the compiler switches),
    so that a PMC summary is tied to the code it profiled and survives a rebuild of the same sources. unit = "quad": over the sources of the
    quad kernel's translation unit alone (build.QUAD_DEPS: quad_kernel.o is built from nothing else), so that a change to the other kernel
    families does not orphan the counters of rollout_quad_kernel"""
    import glob
    from mujoco_mpc_amd import build, capi
    capi.lib()  # (the library must exist: the product path fails loudly without it)
    csrc = os.path.join(ROOT, "mujoco_mpc_amd", "csrc")
    if unit in ("quad", "limb"):   # (likewise the limb kernel's unit: build.LIMB_DEPS)
        files = sorted(os.path.normpath(os.path.join(csrc, f)) for f in (build.QUAD_DEPS if unit == "quad" else build.LIMB_DEPS)) + [os.path.join(ROOT, "mujoco_mpc_amd", "build.py")]
    else:
        files = sorted(glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "generated", "*.h"))
                       + [os.path.join(ROOT, "include", "mjpcx.h"), os.path.join(ROOT, "mujoco_mpc_amd", "build.py")])   # (build.py: the compiler switches)
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def host_cpu_budget():
    """what the process may actually use: CPUs the box shows, CPUs in the affinity mask, and the cgroup CPU quota (containers often show
    every core of the host and are throttled to a fraction of them)"""
    visible = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = visible
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    return dict(visible=visible, affinity=affinity, quota=quota)


def cpu_baseline(task, state, horizon, num_nodes, seconds, n_per_call, mocap=None, interp=None, nominal=None, std=None):
    """Times the CPU oracle (a port, NOT MuJoCo) driven through the reference's fan-out structure (one task per candidate, one physics
    arena per worker thread; sampling/planner.cc:355-393) on the GPU box's host cores, on the SAME kind of candidates the device leg rolls:
    clamp(nominal + N(0, sigma)) around the planner's converged nominal spline (pyoracle.noise_candidates with the planner's noise
    spec), so that both legs do the same solver work per rollout. Reports the whole thread-count probe, the single-thread rate, the
    parallel efficiency of the best count, the cgroup quota, and the rate at the reference's default thread count (testspeed: hardware
    threads - 5, mjpc/testspeed_app.cc:24)."""
    from mujoco_mpc_amd import capi
    from oracle import pyoracle
    pm, pt = task.packed_model(), task.packed()
    budget = host_cpu_budget()
    cores = budget["affinity"]
    dt = task.model.get_number("agent_timestep", task.model.timestep)
    times = np.array([k * (horizon - 1) * dt / (num_nodes - 1) for k in range(num_nodes)])
    sigma = task.model.get_number("sampling_exploration", 0.5) if std is None else std
    nom = np.zeros((num_nodes, task.model.nu)) if nominal is None else np.asarray(nominal, float).reshape(num_nodes, task.model.nu)
    ns = capi.make_noise_spec(seed=0, iteration=1, mode=capi.NOISE_SAMPLING, std0=sigma)
    nodes = pyoracle.noise_candidates(pm, ns, num_nodes, nom, np.arange(1, n_per_call + 1))
    ip = capi.SPLINE_CUBIC if interp is None else interp
    # floating-point operations of the reference algorithm per candidate-step, counted on the first candidates of this very batch by the
    # oracle's operation-counting build (oracle/flopcount.h): the numerator of roofline.fp64 / .fp32
    flops = {}
    n_fl = min(32, n_per_call)
    try:
        with pyoracle.counting_flops(flops):
            pyoracle.rollout_batch(pm, pt, state, 0.0, mocap, n_fl, horizon, num_nodes, ip, times, nodes[:n_fl], num_threads=min(8, max(1, cores)))
        flops["candidates"] = n_fl
        flops["per_candidate_step"] = flops["flop"] / (n_fl * horizon)
    except Exception as ex:  # noqa: BLE001  (the counter must never take the line down)
        flops = {"error": repr(ex)}

    def run(n, threads):
        t0 = time.perf_counter()
        pyoracle.rollout_batch_fast(pm, pt, state, 0.0, mocap, n, horizon, num_nodes, ip, times, nodes[:n], num_threads=threads)
        return n / (time.perf_counter() - t0)

    run(min(64, n_per_call), 1)
    single = max(run(min(128, n_per_call), 1), run(min(128, n_per_call), 1))
    probe = {}
    counts = sorted({max(1, cores // d) for d in (1, 2, 4, 8, 16)} | ({max(1, int(round(budget["quota"])))} if budget["quota"] else set()))
    for threads in counts:
        n = min(n_per_call, 32 * threads)
        probe[threads] = max(run(n, threads), run(n, threads))
    best_threads = max(probe, key=probe.get)
    ref_threads = max(1, budget["visible"] - 5)
    ref_rate = run(min(n_per_call, 32 * min(ref_threads, 256)), ref_threads)
    done, t0 = 0, time.perf_counter()
    while True:
        run(n_per_call, best_threads)
        done += n_per_call
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    value = done / el
    return dict(value=value, unit="rollouts/s", cores=best_threads, kind="port",
                sample=f"{done} rollouts of H={horizon} ({el:.1f} s) through the C oracle's ThreadPool-style fan-out on candidates drawn like the "
                       f"device's (noise {sigma} around the planner's nominal), {best_threads} threads; gcc -O3 -march=native -flto; CPU restatement, "
                       f"not MuJoCo",
                host=budget, single_thread=single, parallel_efficiency=value / (best_threads * single), flops=flops,
                probe={str(k): v for k, v in probe.items()},
                reference_default_threads={"threads": ref_threads, "value": ref_rate,
                                           "note": "testspeed's default: hardware threads - 5 (mjpc/testspeed_app.cc:24)"})


def initial_condition(task_name, task, planner):
    """synthetic initial condition: the task's home keyframe"""
    model = task.model
    home = model.keyframes.get("home")
    qpos = np.array(home["qpos"] if home else model.qpos0, float)
    qvel = np.array(home["qvel"] if home else np.zeros(model.nv), float)
    mocap_pos = mocap_quat = None
    if model.nmocap:  # mocap bodies at their model pose (State::Reset)
        ids = [b for b in range(model.nbody) if model.arrays["body_mocapid"][b] >= 0]
        ids.sort(key=lambda b: model.arrays["body_mocapid"][b])
        mocap_pos = np.array([model.arrays["body_pos"][b] for b in ids], float)
        mocap_quat = np.array([model.arrays["body_quat"][b] for b in ids], float)
    if task_name == "HumanoidTrack":
        # Task::Transition edits the simulation state: first keyframe of the motion, interpolated marker positions
        mode = 9   # Walk
        planner.task_transition_state(0.0, mode, qpos, qvel, mocap_pos.reshape(-1))
        task.transition(0.0, mode)   # the Python mirror keeps the frozen residual state for the cpu_baseline leg
    elif hasattr(task, "transition"):
        planner.task_transition(0.0)
    return qpos, qvel, mocap_pos, mocap_quat


def pmc_path(task_name, precision):
    """the newest committed counter summary of the task (profiles/rNN_pmc_<task>_fp<precision>.json)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_{task_name.lower()}_fp{precision}.json")))
    return found[-1] if found else os.path.join(ROOT, "profiles", f"r06_pmc_{task_name.lower()}_fp{precision}.json")


def pmc_summary(task_name, candidates, horizon, precision):
    """Counter-derived figures of the rollout kernel for THIS build: profiles/rNN_pmc_<task>.json (the newest round's) is written by
    tools/pmc_bench.sh (separate --pmc passes over bench.py's own command, as MI355X_MICROARCH.md prescribes: the counters are of the
    launches this file times) and records the sha256 of the kernel sources it profiled; a summary of any other source state is
    ignored (never a stale lookup)."""
    path = pmc_path(task_name, precision)
    try:
        s = json.load(open(path))
    except (OSError, ValueError):
        return None
    if s.get("candidates") != candidates or s.get("horizon") != horizon:
        return None
    same = s.get("src_sha16") == kernel_source_sha16()
    if not same and "rollout_quad_kernel" in s.get("kernel", ""):  # (its translation unit's own sources decide for the quad kernel)
        same = s.get("unit_src_sha16") == kernel_source_sha16("quad")
    if not same and "rollout_limb_kernel" in s.get("kernel", ""):
        same = s.get("unit_src_sha16") == kernel_source_sha16("limb")
    return s if same else None


def box_probe(device_index=0):
    """Which kind of box this is, in three numbers taken AFTER the timed region (half a second; library kernels, none of this repository's):
    fp64 GEMM rate (rocBLAS, 4096^3), device-to-device copy rate (256 MiB) and the round trip of a one-element kernel with a synchronisation.
    The pool's boxes run the fp64 contact kernels 48-51 ms or 60-77 ms at the same reported clocks: a reader of the line can
    tell from these which kind produced it -- the slow kind shows in the copy rate and the round trip, not in sclk."""
    try:
        import torch
        dev = torch.device("cuda", device_index)
        x = torch.randn(4096, 4096, dtype=torch.float64, device=dev)
        y = torch.randn(4096, 4096, dtype=torch.float64, device=dev)
        torch.mm(x, y); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(4):
            torch.mm(x, y)
        torch.cuda.synchronize(dev)
        gemm = 4 * 2 * 4096 ** 3 / (time.perf_counter() - t0) / 1e12
        a = torch.empty(32 * 2 ** 20, dtype=torch.float64, device=dev)
        b = torch.empty_like(a)
        b.copy_(a); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(8):
            b.copy_(a)
        torch.cuda.synchronize(dev)
        copy = 8 * 2 * a.numel() * 8 / (time.perf_counter() - t0) / 1e9
        one = torch.zeros(1, device=dev)
        one.add_(1); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(200):
            one.add_(1)
            torch.cuda.synchronize(dev)
        rt = (time.perf_counter() - t0) / 200 * 1e6
        return {"dgemm_4096_tflops": gemm, "copy_256mib_gbs": copy, "launch_sync_round_trip_us": rt}
    except Exception as e:  # noqa: BLE001 -- a probe must never take the line down
        return {"error": str(e)}


class ClockSampler:
    """sclk / socket power of the GPU sampled at about 10 Hz from sysfs (hwmon freq1_input, power1_average | power1_input; no
    subprocess on the timed path) while the timed region runs: the headline carries the clock it was measured at. A box without the
    files yields None."""

    def __init__(self, device_index=0):
        import ctypes
        import glob
        # the HIP device's PCI address names its sysfs node (a node shows every GPU of the box, whatever this process may use)
        self.freq = self.power = self.pci = self.mfreq = None
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0:
                self.pci = buf.value.decode().lower()
        except OSError:
            pass
        for card in glob.glob("/sys/class/drm/card[0-9]*/device"):
            if self.pci is None or os.path.basename(os.path.realpath(card)).lower() != self.pci:
                continue
            for h in sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))):
                f = os.path.join(h, "freq1_input")
                if os.path.exists(f):
                    self.freq = f
                f2 = os.path.join(h, "freq2_input")     # (mclk: the boxes of the pool differ by up to 35 % on the same build; memory-side state is the suspect)
                if os.path.exists(f2):
                    self.mfreq = f2
                for name in ("power1_average", "power1_input"):
                    q = os.path.join(h, name)
                    if self.power is None and os.path.exists(q):
                        self.power = q
        self.dpm = {}     # fabric / SoC clock levels (pp_dpm_fclk, pp_dpm_socclk: the line marked '*'), read when the sampling stops
        self._card = None
        for card in glob.glob("/sys/class/drm/card[0-9]*/device"):
            if self.pci is not None and os.path.basename(os.path.realpath(card)).lower() == self.pci:
                self._card = card
        self.samples = []
        self.msamples = []
        self._stop = False
        self._thread = None

    def _read(self, path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        while not self._stop:
            self.samples.append((self._read(self.freq) if self.freq else None, self._read(self.power) if self.power else None))
            if self.mfreq:
                self.msamples.append(self._read(self.mfreq))
            time.sleep(0.1)

    def __enter__(self):
        if self.freq or self.power:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._card:
            for name in ("pp_dpm_fclk", "pp_dpm_socclk"):
                try:
                    for line in open(os.path.join(self._card, name)):
                        if "*" in line:
                            self.dpm[name[7:] + "_mhz"] = float("".join(ch for ch in line.split(":")[1] if ch.isdigit() or ch == "."))
                except (OSError, ValueError, IndexError):
                    pass
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)

    def summary(self):
        f = [a * 1e-6 for a, _ in self.samples if a]
        p = [b * 1e-6 for _, b in self.samples if b]
        if not f and not p:
            return None
        out = {"samples": len(self.samples), "pci": self.pci, "source": "sysfs hwmon of the HIP device's PCI address, 10 Hz over the timed region"}
        if f:
            out.update({"sclk_mhz_min": min(f), "sclk_mhz_mean": sum(f) / len(f), "sclk_mhz_max": max(f)})
        if p:
            out.update({"power_w_mean": sum(p) / len(p), "power_w_max": max(p)})
        mm = [x * 1e-6 for x in self.msamples if x]
        if mm:
            out["mclk_mhz_mean"] = sum(mm) / len(mm)
        out.update(self.dpm)
        return out


def run_config(args, task_name, kind, candidates, horizon, precision, steps, warmup, world, local_rank, group, want_cpu, rank, native=None,
               total=None):
    """one BASELINE config through the C++ planner over the C ABI; returns the fields of a bench line. candidates: this rank's share;
    total: the global batch (default candidates * world)"""
    import torch
    from mujoco_mpc_amd import capi
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.task import load_task

    task = load_task(task_name)
    H = horizon
    total = total or candidates * world
    planner, transport = None, ("rccl" if native is not None else "torch")
    if native is not None:
        # RCCL inside libmjpcx.so; if the communicator cannot be created on ANY rank (mismatched RCCL builds, IPC limits), every
        # rank falls back together to the torch.distributed callbacks -- the line says which transport ran
        try:
            planner = HostPlanner(task, device=local_rank, precision=precision, seed=0, num_trajectory=total, kind=kind,
                                  native_comm=native)
            failed = 0.0
        except Exception as ex:  # noqa: BLE001
            print(f"rank {rank}: native communicator failed ({ex}); falling back to torch.distributed", file=sys.stderr, flush=True)
            failed = 1.0
        if group.max_scalar(failed) > 0:
            if planner is not None:
                planner.close()
            planner, native, transport = None, None, "torch (native communicator failed)"
    if planner is None:
        planner = HostPlanner(task, device=local_rank, precision=precision, seed=0,
                              num_trajectory=total,  # lifts kMaxTrajectory = 128
                              group=group, kind=kind)
    physics_notes = capi.lib().mjpcx_create_error().decode()   # "" or what of the model the device does not reproduce (skipped geom pairs)
    qpos, qvel, mocap_pos, mocap_quat = initial_condition(task_name, task, planner)
    planner.reset(H)
    P = planner.num_spline_points
    planner.set_state(qpos, qvel, 0.0, mocap_pos=mocap_pos, mocap_quat=mocap_quat)

    def fence():
        if group is not None:
            group.barrier()
        planner.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    for _ in range(warmup):
        planner.optimize_policy(H)
    fence()
    planner.timing_reset()
    with ClockSampler(local_rank) as clocks:
        t0 = time.perf_counter()
        for _ in range(steps):
            planner.optimize_policy(H)
        fence()
        elapsed = time.perf_counter() - t0
    main_ms, _ = planner.timing_read_main()
    kernel_ms, launches = planner.timing_read()
    handed_on = planner.quad_stats()
    nominal_nodes = planner.policy()[1]
    if group is not None:
        elapsed = group.max_scalar(elapsed)
    value = total * steps / elapsed
    if rank != 0:
        planner.close()
        return None
    bytes_per_launch = planner.algorithmic_bytes(H, P) * candidates
    avg_kernel_s = main_ms / max(launches, 1) * 1e-3      # the kernel that rolls the batch out (the roofline's kernel)
    avg_rollout_s = kernel_ms / max(launches, 1) * 1e-3   # + the pass over the candidates it handed on / the lane family's sensor stage
    achieved = bytes_per_launch / avg_kernel_s / 1e9
    interp = "cubic" if kind == "sampling" and task_name != "QuadrupedFlat" else None
    label = BASELINE_SIZE.get(task_name, (0, 0, ""))[2]
    out = {
        "value": value, "unit": "rollouts/s", "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "dtype": "f64" if precision == 64 else "f32",
        "config": {"workload": f"{task_name} {'Predictive Sampling' if kind == 'sampling' else 'Cross-Entropy'}, {candidates} candidates/GPU, "
                               f"horizon {H}, {P} spline points, fp{precision} ({label})",
                   "candidates_per_gpu": candidates, "horizon": H, "spline_points": P,
                   "parallelism": f"candidates sharded over {world} rank(s)" + (", exchange over RCCL inside libmjpcx.so" if native is not None else (f", exchange over {transport}" if world > 1 else "")),
                   "kernel": planner.kernel_name,
                   "physics_not_reproduced": physics_notes or None,
                   "rccl_world": planner.comm_info()[1],   # ranks of the library's own RCCL communicator (1: none was created)
                   "host": ("C++ mjpc::GpuSamplingPlanner" if kind == "sampling" else "C++ mjpc::GpuCrossEntropyPlanner") + " over the C ABI"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel_ms": avg_kernel_s * 1e3, "all_rollout_kernels_ms": avg_rollout_s * 1e3, "bytes_per_launch": bytes_per_launch,
                     "handed_on_last_step": {"candidates": handed_on[0], "contact_list_full": handed_on[1], "leg_leg_contact": handed_on[2],
                                             "indefinite_hessian": handed_on[3], "non_finite": handed_on[4], "both_limits": handed_on[5],
                                             "trunk_leg_contact": handed_on[6], "out_of_proof_range": handed_on[7]},
                     "note": "algorithmic bytes (SURVEY 8d) / HIP-event time of the kernel that rolls the batch out (mjpcx_timing_read_main), on "
                             "the context's stream; all_rollout_kernels_ms adds the pass over the candidates it handed to the "
                             "wavefront-per-candidate kernel. The contact models are latency / issue-bound, not HBM-bound (DESIGN.md 4): see `valu`"},
    }
    del interp
    out["gpu_clock"] = clocks.summary()
    if out["gpu_clock"] is not None and not getattr(args, "no_box_probe", False):
        out["gpu_clock"]["box_probe"] = box_probe(local_rank)
    pmc = pmc_summary(task_name, candidates, H, precision)
    if pmc is not None:
        out["roofline"]["traffic"] = pmc.get("hbm_bytes_per_launch")
        # NOT collected in this run: counters need rocprofv3 around the process. The block below is replayed from the committed summary
        # of the same sources (hash-checked above) so that no reader takes it for a measurement of this run
        out["roofline"]["replayed"] = True
        out["roofline"]["replayed_from"] = os.path.relpath(pmc_path(task_name, precision), ROOT)
        t = pmc.get("timed") or {}
        out["roofline"]["traffic_collected_on"] = {
            "launches": f"the {pmc.get('launches', 0) - pmc.get('warmup_launches', 0)} timed launches of `{pmc.get('command')}` (tools/pmc_bench.sh)",
            "kernel_ms_under_kernel_trace": t.get("kernel_ms_under_kernel_trace"), "hbm_bytes_min_max": t.get("hbm_bytes_min_max")}
        if "valu" in pmc:
            out["roofline"]["valu"] = pmc["valu"]
    if want_cpu:
        st = np.concatenate([qpos, qvel])
        cores = os.cpu_count() or 1
        n_cpu = min(max(1024, 32 * cores), candidates) if task_name != "Cartpole" else max(candidates, 64 * cores)
        out["cpu_baseline"] = cpu_baseline(task, st, H, P, args.cpu_seconds, n_cpu,
                                           mocap=None if mocap_pos is None else np.hstack([mocap_pos, mocap_quat]).reshape(-1),
                                           interp=capi.SPLINE_ZERO if task_name == "QuadrupedFlat" or kind == "cross_entropy" else capi.SPLINE_CUBIC,
                                           nominal=nominal_nodes if nominal_nodes.shape[0] == P else None)
        fl = out["cpu_baseline"].get("flops") or {}
        if "per_candidate_step" in fl:
            # the roofline that binds the contact models: the reference algorithm's floating-point operations (as the oracle restates it:
            # dense Jacobian rows, dense Cholesky of the nv x nv Hessian) / the rollout kernel's time, against the vector peak of the dtype
            peak = FP64_VALU_PEAK_TF if precision == 64 else FP32_VALU_PEAK_TF
            tf = fl["per_candidate_step"] * candidates * H / avg_kernel_s / 1e12
            # NOT a utilisation of the device: the numerator is the REFERENCE ALGORITHM's operation count (dense Jacobian rows, dense
            # nv x nv Cholesky, as the oracle restates it), which the device's formulation never executes -- so the field names say
            # "reference-algorithm equivalent" and no fraction of the hardware peak is formed from it. What the device executes is in
            # `executed` below (hardware counters), when a counter summary of these sources exists.
            out["roofline"]["fp64" if precision == 64 else "fp32"] = {
                "reference_algorithm_flop_per_candidate_step": fl["per_candidate_step"], "reference_algorithm_tflops_equivalent": tf,
                "vector_peak_tflops": peak, "unit": "TFLOP/s",
                "counted": {k: fl[k] for k in ("add", "mul", "div", "sqrt", "libm", "candidates")},
                "note": "flop = additions + multiplications + divisions + square roots + other libm calls (one each) executed by the CPU "
                        "oracle's operation-counting build (oracle/flopcount.h) on the first candidates of the batch the device rolls, "
                        "divided by the device kernel's time: the rate a processor running the REFERENCE's dense algorithm would need to "
                        "match this kernel. The device's own formulation (no Jacobian, arrowhead factorisation) executes fewer operations; "
                        "its executed rate is `executed` (from SQ_INSTS_VALU_*_F64 / F32 counters), not this figure"}

    if pmc is not None and pmc.get("executed_flops"):
        # what the device EXECUTES in floating point (hardware counters of the timed launches, replayed like `traffic`): an upper bound of
        # the useful work -- every lane of a wave-instruction is credited (`lane_activity` says how full they were)
        peak = FP64_VALU_PEAK_TF if precision == 64 else FP32_VALU_PEAK_TF
        ex = dict(pmc["executed_flops"])
        if ex.get("flop_per_launch"):
            ex["achieved_tflops"] = ex["flop_per_launch"] / avg_kernel_s / 1e12
            ex["peak"] = peak
            ex["frac"] = ex["achieved_tflops"] / peak
        ex["replayed_from"] = os.path.relpath(pmc_path(task_name, precision), ROOT)
        out["roofline"]["fp64_executed" if precision == 64 else "fp32_executed"] = ex
    planner.close()
    return out


def run_ilqg(local_rank, iterations=6, warmup=2, cpu=True):
    """BASELINE configs[4]: one iLQG iteration on the Quadruped (T = 36, 10 line-search rollouts, forward differences,
    MakeDifferentiable on) through the C++ mjpc::GpuILQGPlanner; beside it the same iteration by the Python mirror of the
    planner on the CPU oracle (tests/oracle_backend.py: oracle/{ilqg,riccati}.c) on the host cores."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.task import load_task
    task = load_task("QuadrupedFlat")
    planner = HostPlanner(task, device=local_rank, precision=64, seed=0, kind="ilqg")
    qpos, qvel, mocap_pos, mocap_quat = initial_condition("QuadrupedFlat", task, planner)
    T = task.planning_steps()
    planner.reset(T)
    planner.set_state(qpos, qvel, 0.0, mocap_pos=mocap_pos, mocap_quat=mocap_quat)
    for _ in range(warmup):
        planner.optimize_policy(T)
    planner.sync()
    t0 = time.perf_counter()
    for _ in range(iterations):
        planner.optimize_policy(T)
    planner.sync()
    gpu_ms = (time.perf_counter() - t0) / iterations * 1e3
    info = planner.ilqg_info()
    planner.close()
    out = {"name": "configs[4] Quadruped iLQG iteration", "value": gpu_ms, "unit": "ms/iteration", "higher_is_better": False,
           "iterations": iterations, "dtype": "f64",
           "config": {"workload": f"QuadrupedFlat iLQG, T = {T}, 10 line-search rollouts, forward differences, derivative_skip 0, fp64",
                      "host": "C++ mjpc::GpuILQGPlanner over the C ABI"},
           "total_return": info["total_return"]}
    # the Riccati backward pass of this shape on its own (its share of the MFMA f64 peak): backward_pass_kernel on a random
    # SPD problem with the config's dimensions -- the part of the iteration that is dense linear algebra
    try:
        from mujoco_mpc_amd import capi
        n_x, n_u = 2 * task.model.nv, task.model.nu
        rng = np.random.default_rng(0)
        A = np.eye(n_x)[None] + 0.1 * rng.normal(size=(T, n_x, n_x))
        B = 0.3 * rng.normal(size=(T, n_x, n_u))

        def spd(k, scale):
            M = rng.normal(size=(T, k, k))
            return scale * (M @ np.transpose(M, (0, 2, 1)) / k + 0.5 * np.eye(k))
        prob = (A, B, rng.normal(size=(T, n_x)), rng.normal(size=(T, n_u)), spd(n_x, 1.0), 0.05 * rng.normal(size=(T, n_x, n_u)), spd(n_u, 0.5),
                rng.uniform(-0.9, 0.9, size=(T, n_u)), np.tile([-1.0, 1.0], (n_u, 1)))
        ctx = capi.Context(task.packed_model(), task.packed(), local_rank, 64)
        ms = [ctx.backward_pass(0.3, 0, 1, *prob)["kernel_ms"] for _ in range(10)]
        ctx.close()
        flops = (T - 1) * 2.0 * (2 * n_x ** 3 + 3 * n_x * n_x * n_u + 2 * n_x * n_u * n_u + n_u ** 3 / 3 + n_x * n_x * n_u)
        kernel_s = float(np.median(ms)) * 1e-3
        out["backward_pass"] = {"n": n_x, "m": n_u, "T": T, "kernel_ms": kernel_s * 1e3,
                                "roofline": {"bound": "mfma", "achieved": flops / kernel_s / 1e12, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                                             "frac": flops / kernel_s / 1e12 / FP64_VALU_PEAK_TF, "traffic": None,
                                             "note": "Riccati flops of the (T - 1) steps / HIP-event time of backward_pass_kernel (v_mfma_f64_16x16x4_f64); "
                                                     "peak = MI355X FP64 matrix = vector peak. One workgroup walks T sequential steps of 36 x 36 "
                                                     "products: latency-bound by construction (DESIGN.md 4.4)"}}
    except Exception as e:  # noqa: BLE001
        out["backward_pass"] = {"error": repr(e)}
    if not cpu:   # (A/B runs of device builds: tools/ab_ilqg.sh)
        return out
    # CPU port of the same iteration
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_backend import OracleContext
        from mujoco_mpc_amd.planners import GpuILQGPlanner, State
        task2 = load_task("QuadrupedFlat")
        task2.transition(0.0)
        threads = max(1, min(16, (os.cpu_count() or 1)))
        from oracle import pyoracle
        with pyoracle.timing_build():   # the -O3 -march=native -flto build of the oracle, as for the rollout legs
            pl = GpuILQGPlanner(backend_factory=lambda tk: OracleContext(tk, threads=threads, differentiable=True))
            pl.initialize(task2.model, task2); pl.allocate(); pl.reset(T)
            st = State(task2.model)
            st.set(qpos, qvel, mocap_pos=mocap_pos, mocap_quat=mocap_quat, time=0.0)
            pl.set_state(st)
            pl.optimize_policy(T)
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                pl.optimize_policy(T)
            cpu_ms = (time.perf_counter() - t0) / n * 1e3
            del pl
        out["cpu_baseline"] = {"value": cpu_ms, "unit": "ms/iteration", "cores": threads, "kind": "port",
                               "sample": f"{n} iterations of the planner on the C oracle (gcc -O3 -march=native -flto): the {T} calls of the "
                                         f"derivative sweep and the line-search rollouts fanned over {threads} worker threads with a physics "
                                         "arena each (oracle/ilqg.c, as the reference's ThreadPool schedules them), the nominal rollout and the "
                                         "Riccati pass on one; the planner loop around them is the Python mirror; CPU restatement, not MuJoCo"}
    except Exception as e:  # the CPU leg must never take the bench line down
        out["cpu_baseline"] = {"error": repr(e)}
    return out


class _OneOfMany:
    """transport of a rank whose peers are not there: every exchange returns the rank's own record (one-GPU emulation of a rank's share)"""
    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def barrier(self):
        pass

    def exchange_best(self, idx, best, nominal, values):
        return idx, best, nominal, values

    def merge_topk(self, idx, ret, k):
        return idx, ret

    def sum_array(self, v):
        return v

    def max_scalar(self, v):
        return v


def emulate_strong(args, total, H, local_rank):
    """--emulate-world W on one GPU: the share of every rank of a W-rank strong-scaling run (total / W candidates at its candidate offset),
    one after the other through the C++ planner, timed like a bench step. The slowest share is what --scaling strong would report per
    step on W GPUs (plus the exchange); printed as one JSON line."""
    from mujoco_mpc_amd.hostplanner import HostPlanner
    from mujoco_mpc_amd.task import load_task
    W = args.emulate_world
    task = load_task(args.task)
    per_rank = []
    kernel = None
    for r in range(W):
        planner = HostPlanner(task, device=local_rank, precision=args.precision, seed=0, num_trajectory=total, kind=args.planner,
                              group=_OneOfMany(r, W))
        qpos, qvel, mocap_pos, mocap_quat = initial_condition(args.task, task, planner)
        planner.reset(H)
        planner.set_state(qpos, qvel, 0.0, mocap_pos=mocap_pos, mocap_quat=mocap_quat)
        for _ in range(args.warmup):
            planner.optimize_policy(H)
        planner.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            planner.optimize_policy(H)
        planner.sync()
        per_rank.append((time.perf_counter() - t0) / args.steps * 1e3)
        kernel = planner.kernel_name
        planner.close()
    slowest = max(per_rank)
    print(json.dumps({"metric": "candidate-trajectory rollouts/sec (fixed horizon)", "emulation": True, "value": total / slowest * 1e3,
                      "unit": "rollouts/s", "n_gpus": 1, "emulated_world": W, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": slowest, "per_rank_ms": per_rank, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                      "dtype": "f64" if args.precision == 64 else "f32", "data": "synthetic",
                      "config": {"workload": f"{args.task}, global batch {total}, horizon {H}: each of {W} ranks' share ({total // W} candidates) timed "
                                             "in turn on ONE GPU, no exchange", "kernel": kernel},
                      "note": "value = global batch / the slowest rank's step time: the throughput an ideal exchange would give W GPUs"}), flush=True)


def run_testspeed(local_rank, plans=100):
    """BASELINE configs[0]: the reference's plumbing case -- testspeed_app's closed loop on Cartpole with 8 Predictive-Sampling candidates
    (simulate + plan synchronously; host/tests/testspeed_app.cc has testspeed_app.cc's flags) -- `plans` plan iterations; beside it the
    CPU port rolling the same 8 candidates out."""
    import re
    import subprocess
    import tempfile
    from mujoco_mpc_amd import mjcf
    from mujoco_mpc_amd.build import build_host
    from mujoco_mpc_amd.task import load_task
    from oracle import pyoracle
    build_host()
    task = load_task("Cartpole")
    d = tempfile.mkdtemp(prefix="mjpx_")
    mjcf.save_blob(task.model, os.path.join(d, "Cartpole.mjpx"))
    spi = 4
    dt = float(task.model.scalars["timestep"])
    total_time = plans * spi * dt
    exe = os.path.join(ROOT, "mujoco_mpc_amd", "host", "build", "testspeed_app")
    env = dict(os.environ, HIP_VISIBLE_DEVICES=str(local_rank))
    out = subprocess.run([exe, "--task=Cartpole", f"--total_time={total_time}", f"--steps_per_planning_iteration={spi}", f"--model_dir={d}",
                          "--candidates=8"], capture_output=True, text=True, timeout=300, env=env)
    if out.returncode != 0:
        raise RuntimeError(out.stdout[-500:] + out.stderr[-500:])
    m1 = re.search(r"(\d+) plan iterations of (\d+) candidates x (\d+) steps\): ([0-9.]+) s", out.stdout)
    m2 = re.search(r"Mean plan iteration: ([0-9.]+) us", out.stdout)
    m3 = re.search(r"Average cost per step \(lower is better\): ([-0-9.eE+]+)", out.stdout)
    nplans, ncand, steps, wall = int(m1.group(1)), int(m1.group(2)), int(m1.group(3)), float(m1.group(4))
    plan_us = float(m2.group(1))
    line = {"name": "configs[0] testspeed closed loop", "metric": "candidate-trajectory rollouts/sec (fixed horizon)",
            "value": ncand / (plan_us * 1e-6), "unit": "rollouts/s", "higher_is_better": True, "dtype": "f64",
            "plan_iterations": nplans, "mean_plan_iteration_us": plan_us, "closed_loop_wall_s": wall, "average_cost": float(m3.group(1)),
            "config": {"workload": f"Cartpole Predictive Sampling, {ncand} candidates, horizon {steps} steps, testspeed_app closed loop "
                                   "(BASELINE.json configs[0]: the reference's own CPU-runnable case)",
                       "host": "host/tests/testspeed_app.cc -> mjpc::SynchronousPlanningCost (simulation step and planner both on the device)"},
            "note": "latency-bound by construction (8 candidates = 8 lanes of one wavefront, one launch + one sync per plan iteration): the "
                    "plumbing check BASELINE.json lists first, not a throughput case"}
    # CPU port: the same number of candidates and steps through the oracle's thread pool
    pm, pt = task.packed_model(), task.packed()
    P = 3
    times = np.arange(P) * ((steps - 1) * dt / (P - 1))
    nodes = np.random.default_rng(0).normal(0, 0.1, (ncand, P, task.model.nu))
    st = np.zeros(task.model.nq + task.model.nv)
    budget = host_cpu_budget()
    threads = max(1, min(ncand, int(budget["quota"] or budget["affinity"])))
    pyoracle.rollout_batch(pm, pt, st, 0.0, None, ncand, steps, P, 0, times, nodes, num_threads=threads)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 2.0:
        pyoracle.rollout_batch(pm, pt, st, 0.0, None, ncand, steps, P, 0, times, nodes, num_threads=threads)
        reps += 1
    cpu = reps * ncand / (time.perf_counter() - t0)
    line["cpu_baseline"] = {"value": cpu, "unit": "rollouts/s", "cores": threads, "kind": "port",
                            "sample": f"{reps} batches of {ncand} rollouts x {steps} steps through oracle/ (thread pool of {threads}); CPU restatement, not MuJoCo"}
    return line


def dry_run(args, rank, world):
    """The --gpus N launch path with the device taken out (tests/test_distributed_gloo.py runs it at WORLD_SIZE = 2): what can
    be wrong the first time an 8-GPU node appears -- env parsing, rendezvous, the 128-byte communicator id reaching every rank,
    candidate ranges that tile [0, N) -- is exercised here; the RCCL calls themselves are covered at world = 1 on the GPU."""
    import torch
    import torch.distributed as dist
    from mujoco_mpc_amd.distributed import RankGroup
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("MJPC_BENCH_DRY_RUN_FAIL_RANK") == str(rank):   # (tests: a rank that dies before the rendezvous)
        sys.exit(3)
    if world > 1:
        dist.init_process_group(backend="gloo")
    group = RankGroup(dist, torch.device("cpu")) if world > 1 else None
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(hashlib.sha256(b"dry-run communicator id").digest() * 4), dtype=torch.uint8))
    if world > 1:
        dist.broadcast(uid, src=0)
    assert bytes(uid.numpy().tobytes()) == hashlib.sha256(b"dry-run communicator id").digest() * 4
    n0, h0, _ = BASELINE_SIZE.get(args.task, (4096, 128, ""))
    candidates = args.candidates or n0
    total = candidates if args.scaling == "strong" else candidates * world
    q, r = divmod(total, world)                      # the planners' split: the first N % world ranks hold one more candidate
    begin = rank * q + min(rank, r)
    count = q + (1 if rank < r else 0)
    if group is not None:
        assert all(group.owner_of(g, total) == rank for g in (begin, begin + count - 1))
        group.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1) * args.steps)       # rank-dependent "work": the reported time must be the slowest rank's
    if group is not None:
        group.barrier()
    elapsed = time.perf_counter() - t0
    if group is not None:
        elapsed = group.max_scalar(elapsed)
        covered = group.max_scalar(float(begin + count))
        assert covered == total
    if rank == 0:
        assert elapsed >= 0.01 * world * args.steps
        print(json.dumps({"metric": "candidate-trajectory rollouts/sec (fixed horizon)", "value": None, "unit": "rollouts/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                          "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "dry_run": True,
                          "config": {"workload": f"{args.task} launcher dry run", "candidates_per_gpu": candidates,
                                     "ranges": "contiguous, first N % world ranks +1"}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start one copy of this command per GPU (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR = 127.0.0.1 / a free MASTER_PORT in the environment, as torch.distributed.run would set them), pass rank 0's
    standard output through (the one JSON line), and exit non-zero -- after stopping the others -- as soon as any rank dies.
    MJPC_BENCH_LAUNCH_TIMEOUT_S bounds the whole run (default 1800 s)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + float(os.environ.get("MJPC_BENCH_LAUNCH_TIMEOUT_S", "1800"))
    rc = 0
    live = list(range(n))
    while live and rc == 0:
        for r in list(live):
            code = procs[r].poll()
            if code is not None:
                live.remove(r)
                if code != 0:
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
                    rc = code if code > 0 else 1
        if time.monotonic() > deadline:
            print("bench.py: launch timed out", file=sys.stderr)
            rc = 124
        time.sleep(0.05)
    for r in live:                      # a rank died or the deadline passed: stop the ranks this process started (exact PIDs)
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    sys.exit(rc)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)      # plain `python bench.py --gpus N`: one child per GPU, this process only supervises
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, rank, world)
    import torch

    group = None
    native = None
    if world > 1:
        # torch.distributed only rendezvouses the ranks (barriers, the max over ranks of the elapsed time, and shipping the
        # RCCL unique id); the per-step candidate exchange runs inside libmjpcx.so on its own RCCL communicator
        import torch.distributed as dist
        from mujoco_mpc_amd.distributed import RankGroup
        from mujoco_mpc_amd.hostplanner import comm_unique_id
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        group = RankGroup(dist, torch.device("cuda", local_rank))
        if os.environ.get("MJPC_BENCH_TRANSPORT", "rccl") != "torch":  # ("torch": the exchange through torch.distributed callbacks)
            uid = torch.zeros(128, dtype=torch.uint8, device=torch.device("cuda", local_rank))
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, src=0)
            native = (bytes(uid.cpu().numpy().tobytes()), rank, world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    n0, h0, _ = BASELINE_SIZE.get(args.task, (4096, 128, ""))
    candidates = args.candidates or n0
    H = args.horizon or h0
    total = None
    if args.scaling == "strong":   # the global batch is fixed; a rank's share is total / world (the first total % world ranks hold one more)
        total = candidates
        candidates = total // world + (1 if rank < total % world else 0)
    if args.emulate_world > 1 and world == 1:
        return emulate_strong(args, candidates, H, local_rank)
    main_line = run_config(args, args.task, args.planner, candidates, H, args.precision, args.steps, args.warmup, world, local_rank,
                           group, want_cpu=not args.no_cpu_baseline and world == 1, rank=rank, native=native, total=total)
    if rank == 0:
        out = {"metric": "candidate-trajectory rollouts/sec (fixed horizon)", "value": main_line["value"], "unit": "rollouts/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_line["ms_per_step"],
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": main_line["dtype"], "data": "synthetic",
               "config": main_line["config"], "roofline": main_line["roofline"], "gpu_clock": main_line.get("gpu_clock")}
        if "cpu_baseline" in main_line:
            out["cpu_baseline"] = main_line["cpu_baseline"]
            cb = main_line["cpu_baseline"]
            out["target_check"] = {"gpu_over_cpu": main_line["value"] / cb["value"], "north_star_target": 64.0,
                                   "gpu_over_one_cpu_thread": main_line["value"] / cb["single_thread"],
                                   "gpu_over_cpu_at_reference_default_threads": main_line["value"] / cb["reference_default_threads"]["value"],
                                   "note": "GPU rollouts/s over the CPU port's at its best thread count (a port, not MuJoCo); the box's cgroup quota "
                                           "and the whole probe are in cpu_baseline.host / .probe"}
        default_run = (args.task == "QuadrupedFlat" and args.planner == "sampling" and not args.candidates and not args.horizon
                       and args.precision == 64)
        if world == 1 and default_run and not args.no_extra:
            extra = []
            for name, task_name, kind, prec, steps in (("configs[1]", "Cartpole", "sampling", 64, 50),
                                                       ("configs[2] (Cross-Entropy planner)", "QuadrupedFlat", "cross_entropy", 64, 5),
                                                       ("configs[3] (one GPU's share)", "HumanoidTrack", "sampling", 32, 10)):
                n, h, _ = BASELINE_SIZE[task_name]
                try:
                    e = run_config(args, task_name, kind, n, h, prec, steps, 2, 1, local_rank, None,
                                   want_cpu=(task_name != "QuadrupedFlat") and not args.no_cpu_baseline, rank=0)
                    e["name"] = name
                    e["metric"] = "candidate-trajectory rollouts/sec (fixed horizon)"
                except Exception as ex:
                    e = {"name": name, "error": repr(ex)}
                extra.append(e)
            try:
                extra.append(run_testspeed(local_rank))
            except Exception as ex:
                extra.append({"name": "configs[0] testspeed closed loop", "error": repr(ex)})
            try:
                extra.append(run_ilqg(local_rank))
            except Exception as ex:
                extra.append({"name": "configs[4] Quadruped iLQG iteration", "error": repr(ex)})
            out["extra"] = extra
        print(json.dumps(out), flush=True)
    if group is not None:
        group.barrier()
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
