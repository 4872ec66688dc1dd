#!/usr/bin/env python3
"""A variant of libmjpcx.so for same-box A/B runs (tools/ab_quad.sh): the quad kernel's translation unit compiled with extra flags, linked with
the in-tree objects of the other units. python tools/build_variant.py <name> [flags ...] [--root <tree>] -> mujoco_mpc_amd/libmjpcx_<name>.so"""
import os
import subprocess
import sys

args = sys.argv[1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src_root = root
if "--root" in args:
    i = args.index("--root")
    src_root = args[i + 1]
    del args[i:i + 2]
name, flags = args[0], args[1:]
sys.path.insert(0, src_root)
from mujoco_mpc_amd import build  # noqa: E402

pkg = os.path.join(src_root, "mujoco_mpc_amd")
obj = os.path.join("/tmp", f"quad_kernel_{name}.o")
common = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
subprocess.check_call(common + build.PRESSURE_QUAD + flags + ["-c", os.path.join(pkg, "csrc", "quad_kernel.hip"), "-o", obj])
others = [os.path.join(pkg, "build", os.path.splitext(s)[0] + ".o") for s, _ in build.SOURCES if s != "quad_kernel.hip"]
out = os.path.join(root, "mujoco_mpc_amd", f"libmjpcx_{name}.so")
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + others + [obj, "-o", out])
print(out)
