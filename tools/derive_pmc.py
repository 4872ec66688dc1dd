#!/usr/bin/env python3
"""gpurun_out/pmc/summary.json (tools/pmc_rollout.sh, tools/summarize_pmc.py) -> the per-build counter summary bench.py reads
(profiles/r04_pmc_<task>_fp<prec>.json): python tools/derive_pmc.py <summary.json> <task> <candidates> <horizon> <precision> <out.json>
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB per dispatch; the x2 is MI355X_MICROARCH.md's gfx950 correction for wide reads).
The summary is tied to the code it profiled by the sha256 of the kernel sources (bench.kernel_source_sha16); bench.py ignores it
for any other source state."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, task, N, H, prec, dst = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
summary = json.load(open(src))
# the first pass of the registered kernel (or the generic wave kernel) carries the whole batch
main = max(summary.items(), key=lambda kv: kv[1].get("SQ_WAVE_CYCLES", 0))
name, c = main
sys.path.insert(0, ROOT)
from mujoco_mpc_amd import capi  # noqa: E402
from mujoco_mpc_amd.task import load_task  # noqa: E402
t = load_task(task)
P = int(t.model.get_number("sampling_spline_points", 10))
m = t.packed_model().struct
nr, ntr = t.packed().struct.num_residual, t.packed().struct.num_trace
wsz = 8 if prec == 64 else 4
per_rollout = wsz * (H * (m.nq + m.nv + m.nu + 1 + nr + 3 * ntr + 1) + P * m.nu + P + 2)  # == mjpcx_algorithmic_bytes
# per wavefront and step: the quad kernel packs 16 candidates into a wavefront, the lane kernels 64, the others take one each
per_wave = 16 if "rollout_quad" in name else (64 if "lane" in name else 1)
waves_steps = (N + per_wave - 1) // per_wave * H
out = dict(c)
out.update({
    "kernel": name, "task": task, "candidates": N, "horizon": H, "precision": prec,
    "src_sha16": __import__("bench").kernel_source_sha16(),
    "unit_src_sha16": __import__("bench").kernel_source_sha16("quad") if "rollout_quad_kernel" in name else None,
    "hbm_bytes_per_launch": (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024.0,
    "algorithmic_bytes_per_launch": per_rollout * N,
    "valu": {
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
    "other_kernels": {k: v for k, v in summary.items() if k != name},
})
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "src_sha16", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "valu")}, indent=1))
