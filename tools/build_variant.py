#!/usr/bin/env python3
"""A variant of libmjpcx.so for same-box A/B runs (tools/ab_quad.sh, tools/ab_lib.sh): ONE translation unit compiled with other flags, linked with
the in-tree objects of the other units.
    python tools/build_variant.py <name> [--unit quad_kernel|limb_kernel|wave32|mjpcx|ilqg_wave] [--csrc <dir>] [--bare] [flags ...] -> mujoco_mpc_amd/libmjpcx_<name>.so
The unit's own switches of build.py are kept unless --bare is given (then only the flags on the command line are used)."""
import os
import subprocess
import sys

args = sys.argv[1:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
unit, bare = "quad_kernel", False
if "--unit" in args:
    i = args.index("--unit")
    unit = args[i + 1]
    del args[i:i + 2]
csrc = None
if "--csrc" in args:   # the unit's sources from a copy of csrc/ (an earlier or edited state of the headers, A/B'd against the tree's)
    i = args.index("--csrc")
    csrc = args[i + 1]
    del args[i:i + 2]
if "--bare" in args:
    args.remove("--bare")
    bare = True
name, flags = args[0], args[1:]
sys.path.insert(0, root)
from mujoco_mpc_amd import build  # noqa: E402

pkg = os.path.join(root, "mujoco_mpc_amd")
own = dict((os.path.splitext(s)[0], f) for s, f in build.SOURCES)[unit]
obj = os.path.join("/tmp", f"{unit}_{name}.o")
common = [build._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
subprocess.check_call(common + ([] if bare else own) + flags + ["-c", os.path.join(csrc or os.path.join(pkg, "csrc"), unit + ".hip"), "-o", obj])
others = [os.path.join(pkg, "build", os.path.splitext(s)[0] + ".o") for s, _ in build.SOURCES if os.path.splitext(s)[0] != unit]
out = os.path.join(pkg, f"libmjpcx_{name}.so")
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + others + [obj, "-o", out])
print(out)
