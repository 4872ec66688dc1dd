#!/usr/bin/env python3
"""gpurun_out/pmc_<task>/p0..p7 (tools/pmc_bench.sh: one --kernel-trace pass and seven --pmc passes over bench.py's own command) ->
the per-build counter summary bench.py reads (profiles/rNN_pmc_<task>_fp<prec>.json):
    python tools/pmc_summary.py <dir> <task> <precision> <steps> <warmup> <out.json>
Every figure is kept PER LAUNCH of the task's rollout kernel, in launch order (warm-up launches first), so that the easy first
launches and the timed ones are separable; `timed` averages the launches bench.py times (index >= warmup).
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB per dispatch; the x2 is MI355X_MICROARCH.md's gfx950 correction for wide reads).
The summary is tied to the code it profiled by the sha256 of the kernel sources (bench.kernel_source_sha16); bench.py ignores it
for any other source state. Generated, never edited by hand."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, task, prec, steps, warm, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mujoco_mpc_amd.task import load_task  # noqa: E402


def main_kernel(names):
    """the kernel that rolls the batch out: the rollout kernel with the largest share of the dispatches' time / cycles"""
    best = max(names.items(), key=lambda kv: kv[1])
    return best[0]


# ---- pass 0: durations per launch
trace = {}
for f in glob.glob(os.path.join(d, "p0", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "rollout_" not in k:
            continue
        trace.setdefault(k, []).append((int(row["Dispatch_Id"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6))
if not trace:
    raise SystemExit("no rollout kernel in the kernel trace of " + d)
name = main_kernel({k: sum(ms for _, ms in v) for k, v in trace.items()})
launch_ms = [ms for _, ms in sorted(trace[name])]
# the launches of the batch: a registered kernel may add a short pass over overflowed / handed-on candidates under the same or another
# name; the batch's launch is the long one of each plan step -- keep the launches above a tenth of the longest
cut = 0.1 * max(launch_ms)
keep = [i for i, ms in enumerate(launch_ms) if ms > cut]
launch_ms = [launch_ms[i] for i in keep]

# ---- passes 1..7: counters per launch
per = {}
for p in range(1, 8):
    for f in glob.glob(os.path.join(d, f"p{p}", "**", "*counter_collection.csv"), recursive=True):
        acc, dur = {}, {}
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].split("(")[0] != name:
                continue
            did = int(row["Dispatch_Id"])
            acc.setdefault(row["Counter_Name"], {}).setdefault(did, 0.0)
            acc[row["Counter_Name"]][did] += float(row["Counter_Value"])
            dur[did] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
        ids = sorted(dur)
        big = [i for i in ids if dur[i] > 0.1 * max(dur.values())]
        for cname, v in acc.items():
            per[cname] = [v.get(i, 0.0) for i in big]
        per[f"_pass{p}_ms"] = [dur[i] for i in big]
n = len(launch_ms)
for k, v in per.items():
    if len(v) != n:
        print(f"warning: {k} has {len(v)} launches, the trace {n}", file=sys.stderr)


def timed(v):
    w = v[warm:] if len(v) > warm else v
    return sum(w) / len(w)


t = load_task(task)
P = int(t.model.get_number("sampling_spline_points", 10))
N, H, _ = bench.BASELINE_SIZE[task]
m = t.packed_model().struct
nr, ntr = t.packed().struct.num_residual, t.packed().struct.num_trace
wsz = 8 if prec == 64 else 4
per_rollout = wsz * (H * (m.nq + m.nv + m.nu + 1 + nr + 3 * ntr + 1) + P * m.nu + P + 2)  # == mjpcx_algorithmic_bytes
def limb_cpw(n):   # (limb_kernel.hip pick_cpw: one wavefront per SIMD first, sixteen candidates per wavefront from 16384 on)
    c = 1
    while c < 16 and (n + c - 1) // c > 1024:
        c *= 2
    return c


per_wave = 16 if "rollout_quad" in name else (limb_cpw(N) if "rollout_limb" in name else (64 if "lane" in name else 1))
waves_steps = (N + per_wave - 1) // per_wave * H
c = {k: timed(v) for k, v in per.items() if not k.startswith("_")}
hbm = [(2 * f + w) * 1024.0 for f, w in zip(per.get("FETCH_SIZE", []), per.get("WRITE_SIZE", []))]
out = {
    "kernel": name, "task": task, "candidates": N, "horizon": H, "precision": prec,
    "command": f"bench.py --task {task} --precision {prec} --steps {steps} --warmup {warm} --no-extra --no-cpu-baseline",
    "src_sha16": bench.kernel_source_sha16(),
    "unit_src_sha16": bench.kernel_source_sha16("quad") if "rollout_quad_kernel" in name else (bench.kernel_source_sha16("limb") if "rollout_limb_kernel" in name else None),
    "launches": n, "warmup_launches": warm,
    "launch_ms_under_kernel_trace": launch_ms,
    "launch_ms_under_counters": {k[1:]: v for k, v in per.items() if k.startswith("_")},
    "hbm_bytes_by_launch": hbm,
    "valu_insts_by_launch": per.get("SQ_INSTS_VALU"),
    "wave_cycles_by_launch": per.get("SQ_WAVE_CYCLES"),
    "timed": {
        "kernel_ms_under_kernel_trace": timed(launch_ms),
        "hbm_bytes_per_launch": timed(hbm) if hbm else None,
        "hbm_bytes_min_max": [min(hbm[warm:]), max(hbm[warm:])] if len(hbm) > warm else None,
        "counters": c,
    },
    "hbm_bytes_per_launch": timed(hbm) if hbm else None,
    "algorithmic_bytes_per_launch": per_rollout * N,
    "valu": {
        "collected_on": f"the {n - warm} timed launches of bench.py's own command (launch index >= {warm}); their mean duration under "
                        f"--kernel-trace {timed(launch_ms):.2f} ms",
        "per_wavefront_step": {k: c[f"SQ_INSTS_{k}"] / waves_steps for k in ("VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR", "SMEM") if f"SQ_INSTS_{k}" in c},
        "wave_cycles_per_step": 4 * c["SQ_WAVE_CYCLES"] / waves_steps if "SQ_WAVE_CYCLES" in c else None,
        "wait_any_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in c else None,
        "issue_stall_frac": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if "SQ_WAIT_INST_ANY" in c else None,
        "active_valu_frac": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if "SQ_ACTIVE_INST_VALU" in c else None,
        "active_inst_any_frac": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"] if "SQ_ACTIVE_INST_ANY" in c else None,
        "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else None,
        "icache_hit_frac": c["SQC_ICACHE_HITS"] / c["SQC_ICACHE_REQ"] if c.get("SQC_ICACHE_REQ") else None,
        "candidates_per_wavefront": per_wave,
    },
}
# ---- what the kernel executes in floating point (pass 6 / 7): wave-level instruction counts by class of the working precision. A
# wave-instruction is credited with all 64 lanes, so flop_per_launch is an UPPER bound of the useful work (lanes masked off by EXEC, the
# trunk's arithmetic replicated in the lanes of a candidate, selects and moves are not separated); lane_activity says how full the VALU
# instructions were (SQ_THREAD_CYCLES_VALU: active lanes summed over the VALU cycles, against 64 x the wave-level VALU cycles)
F = f"F{prec}"
if f"SQ_INSTS_VALU_FMA_{F}" in c:
    fma, add, mul, tr = (c.get(f"SQ_INSTS_VALU_{k}_{F}", 0.0) for k in ("FMA", "ADD", "MUL", "TRANS"))
    out["executed_flops"] = {
        "wave_instructions": {"fma": fma, "add": add, "mul": mul, "trans": tr, "all_valu": c.get("SQ_INSTS_VALU"),
                              "int32": c.get("SQ_INSTS_VALU_INT32"), "cvt": c.get("SQ_INSTS_VALU_CVT")},
        "flop_per_launch": (2 * fma + add + mul + tr) * 64,
        "fp_share_of_valu": (fma + add + mul + tr) / c["SQ_INSTS_VALU"] if c.get("SQ_INSTS_VALU") else None,
        "hardware_flops_counter": c.get(f"SQ_INSTS_VALU_FLOPS_FP{prec}"), "hardware_flops_trans_counter": c.get(f"SQ_INSTS_VALU_FLOPS_FP{prec}_TRANS"),
        "thread_cycles_valu": c.get("SQ_THREAD_CYCLES_VALU"),
        "lane_activity": (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])) if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU") else None,
        "note": "flop_per_launch = (2 FMA + ADD + MUL + TRANS wave-instructions of the working precision) x 64 lanes, mean of the timed launches: "
                "an upper bound of the useful floating-point work (every lane credited; see lane_activity). Counters: SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_" + F,
    }
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "src_sha16", "launches", "launch_ms_under_kernel_trace", "hbm_bytes_by_launch", "timed", "valu", "executed_flops") if k in out}, indent=1))
