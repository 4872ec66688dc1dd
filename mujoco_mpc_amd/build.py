"""In-tree native build: hipcc cross-compiles the gfx950 C-ABI library (no GPU needed)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB = os.path.join(PKG_DIR, "libmjpcx.so")
# (source, extra flags). lane_static.hip holds the instantiations specialised for compile-time model
# constants; its flags let exact-zero arithmetic fold (see the file header).
# Register-pressure switches for the units whose kernels live at the edge of the register file (the quad kernel: 512 registers and
# 68 GB of spill traffic per launch; the tree kernels in fp32 and fp64: 256). None of them changes what is computed, only what the optimiser hoists,
# merges or if-converts -- each of those moves lengthens live ranges: loop-invariant code motion (machine level and promotion), sinking
# of common code out of branches, speculation of branch bodies into selects, SLP pairing of scalar loads -- and the scheduler is told to
# weigh occupancy / pressure over latency. Measured on MI355X, same box, back to back: quad kernel 67.5 -> 62.1 ms,
# Humanoid fp32 143.4 -> 155.9 k rollouts/s, the fp64 tree kernels of mjpcx.hip +5 % (A1 at N = 2048: 35.7 -> 33.9 ms; Humanoid fp64 65 -> 69 k;
# its lane kernels and the Riccati pass unchanged); the iLQG unit got slower with them (19.2 -> 20.0 ms) and keeps the defaults.
PRESSURE = ["-fno-slp-vectorize", "-mllvm", "-disable-machine-licm", "-mllvm", "-disable-licm-promotion", "-mllvm", "-simplifycfg-sink-common=false",
            "-mllvm", "-phi-node-folding-threshold=0", "-mllvm", "-amdgpu-schedule-metric-bias=100"]
# the quad unit after round 4's restructuring (line search inlined, no memory-resident Hessian blocks): the scheduler bias and the SLP switch
# no longer pay there (same box, back to back: 58.0 ms with all six, 57.5 with these four; without the LICM pair 59.4, without the CFG pair 59.1)
# -ffp-contract=on (round 5): a*b+c contracts only where the source writes it in one expression, instead of wherever inlining happens to bring
# a product and a sum together -- the two instantiations of the line search (with / without contacts beyond the register slots, chosen per
# WAVEFRONT) then round alike, so a candidate's result does not depend on its wavefront's other candidates (tests/test_gpu_quad.py::
# test_results_do_not_depend_on_the_candidates_per_wavefront saw 5e-16 otherwise), and the launch is 2 % faster (57.3 -> 56.1 ms, same box)
PRESSURE_QUAD = ["-mllvm", "-disable-machine-licm", "-mllvm", "-disable-licm-promotion", "-mllvm", "-simplifycfg-sink-common=false",
                 "-mllvm", "-phi-node-folding-threshold=0", "-ffp-contract=on", "-fno-slp-vectorize"]
# what each group is worth on the round-5 kernel (gait steps of the bench, same box, back to back; 51.1 ms with the set): without the LICM pair 53.5,
# without the CFG pair 51.9, without both (-ffp-contract=on alone) 53.3; + -fno-slp-vectorize 50.8 (twice); + the scheduler bias 51.1 (none);
# -amdgpu-sched-strategy=max-ilp 51.4
SOURCES = [("mjpcx.hip", PRESSURE), ("ilqg_wave.hip", []), ("wave32.hip", PRESSURE), ("lane_static.hip", ["-fno-signed-zeros", "-ffinite-math-only"]),
           ("quad_kernel.hip", PRESSURE_QUAD), ("limb_kernel.hip", [])]
# headers only the quad kernel's translation unit includes / the headers that unit needs (so that a change of the quad step does not
# re-compile the wavefront-per-candidate kernels, and vice versa)
QUAD_ONLY = ["quad_step.h", "quad_kernel.h", "quad_model.h", "limb_step.h", "limb_kernel.h", "limb_model.h"]
# likewise the limb kernel's unit (the Humanoid of configs[3])
LIMB_DEPS = ["limb_step.h", "limb_kernel.h", "limb_model.h", "limb_abi.h", "limb_launch.h", "limb_kernel.hip", "pair_cull.h", "solid_pairs.h",
             os.path.join("..", "..", "include", "mjpcx.h")]
QUAD_DEPS = ["quad_step.h", "quad_kernel.h", "quad_model.h", "quad_abi.h", "quad_launch.h", "quad_kernel.hip", "solid_pairs.h", "pair_cull.h",
             os.path.join("..", "..", "include", "mjpcx.h")]
HEADERS = ["device_common.h", "rollout_lane.h", "lane_registry.h", os.path.join("generated", "static_models.h"),
           os.path.join("..", "..", "include", "mjpcx.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmjpcx.so)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    from . import codegen
    codegen.main([])  # regenerate generated/static_models.h if the model XMLs changed
    import glob
    # every header of csrc/ is a dependency of both translation units (the wave_*.h / ilqg_*.h files are included by mjpcx.hip)
    deps = sorted(set([os.path.join(CSRC, s) for s, _ in SOURCES] + [os.path.join(CSRC, h) for h in HEADERS] +
                      glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "generated", "*.h"))))
    deps.append(os.path.abspath(__file__))
    if not force and not _stale(LIB, deps):
        return LIB
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    common = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
    objs, procs = [], []
    for src, flags in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if src == "quad_kernel.hip":
            mine = [os.path.join(CSRC, h) for h in QUAD_DEPS]
        elif src == "limb_kernel.hip":
            mine = [os.path.join(CSRC, h) for h in LIMB_DEPS]
        else:
            mine = [d for d in deps if os.path.basename(d) not in QUAD_ONLY and os.path.basename(d) not in ("quad_kernel.hip", "limb_kernel.hip")]
        mine = mine + [os.path.abspath(__file__)]  # (the flags are in this file)
        if force or _stale(obj, mine):
            cmd = common + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_host(verbose=False):
    """C++ host layer (namespace mjpc over the C ABI) + its test programs: g++, links libmjpcx.so."""
    build_native()
    host = os.path.join(PKG_DIR, "host")
    subprocess.check_call(["make", "-C", host, "-s", "-j8"], stdout=None if verbose else subprocess.DEVNULL)
    return os.path.join(host, "build", "libmjpc_host.so")
