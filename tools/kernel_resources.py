#!/usr/bin/env python3
"""Code-object metadata of every kernel of libmjpcx.so (registers, spills, private segment, static LDS), from the objects of the in-tree
build: python tools/kernel_resources.py [out.json]. The numbers the register-pressure work iterates on."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
out = {}
for obj in sorted(os.listdir(os.path.join(ROOT, "mujoco_mpc_amd", "build"))):
    if not obj.endswith(".o"):
        continue
    path = os.path.join(ROOT, "mujoco_mpc_amd", "build", obj)
    with tempfile.TemporaryDirectory() as td:
        co = os.path.join(td, "k.co")
        fb = os.path.join(td, "fb.bin")   # the device code sits in the host object's .hip_fatbin section, as an offload bundle
        if subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fb}", path], capture_output=True).returncode != 0:
            continue
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fb}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")

    def emit(rec):
        if rec and "name" in rec:
            name = subprocess.run(["c++filt", rec["name"]], capture_output=True, text=True).stdout.strip()
            out.setdefault(obj, {})[re.sub(r"\(.*", "", name)[:120]] = {k: rec[k] for k in KEYS if k in rec}

    cur = None
    for line in notes.splitlines():
        m = re.match(r"\s+- \.agpr_count:\s+(\d+)", line)
        if m:
            emit(cur)
            cur = {"agpr_count": int(m.group(1))}
            continue
        if cur is None:
            continue
        m = re.match(r"    \.(\w+):\s+(\S+)\s*$", line)   # (the kernel's own keys: four spaces; its arguments' are indented deeper)
        if m and (m.group(1) in KEYS or m.group(1) == "name"):
            cur[m.group(1)] = m.group(2) if m.group(1) == "name" else int(m.group(2))
    emit(cur)
text = json.dumps(out, indent=1, sort_keys=True)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text + "\n")
for obj, ks in out.items():
    for name, r in sorted(ks.items(), key=lambda kv: -kv[1].get("private_segment_fixed_size", 0))[:6]:
        print(f"{obj:18s} {name[:70]:70s} vgpr {r.get('vgpr_count')} agpr {r.get('agpr_count')} vspill {r.get('vgpr_spill_count')} sspill {r.get('sgpr_spill_count')} "
              f"scratch {r.get('private_segment_fixed_size')} lds {r.get('group_segment_fixed_size')}")
