#!/usr/bin/env python3
"""Pins the repo's restated model files to the reference's own model patches (tests/test_model_provenance.py).

The reference does not vendor its robot models: it holds `*.xml.patch` files that turn upstream models (dm_control suite, MuJoCo
Menagerie) into the ones its tasks load (mjpc/tasks/CMakeLists.txt applies them at build time). Every `+` line and every context line of
a patch is a line of the patched model, so every element on those lines must be an element of this repo's restatement of that model.
This script reads the patches from /root/reference and writes, per model, the SHA-1 of each such element (tag + attributes, visual-only
attributes and elements left out) to tests/golden/model_provenance.json: hashes, not the lines, so that no reference text is copied.
The test recomputes the same hashes from the repo's XML and checks that the fixture's are all there.

    python tools/make_model_provenance.py            # rewrite the fixture (needs /root/reference)
"""
import hashlib
import json
import os
import re
import sys
import xml.etree.ElementTree as ET

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/mjpc/tasks"
# (patch in the reference, the repo's restatement of the patched model)
MODELS = {
    "quadruped": ("quadruped/a1.xml.patch", "mujoco_mpc_amd/models/quadruped/a1_modified.xml"),
    "cartpole": ("cartpole/cartpole.xml.patch", "mujoco_mpc_amd/models/cartpole/cartpole.xml"),
    "humanoid": ("humanoid/humanoid.xml.patch", "mujoco_mpc_amd/models/humanoid/humanoid_modified.xml"),
    "particle": ("particle/particle.xml.patch", "mujoco_mpc_amd/models/particle/particle.xml"),
}
# elements / attributes that never reach mjModel's dynamics, collision or sensor fields (rendering only)
VISUAL_TAGS = {"light", "camera", "material", "texture", "mesh", "skin", "visual", "global", "quality", "headlight", "map", "scale", "rgba",
               "statistic", "include", "asset", "hfield"}
VISUAL_ATTRS = {"material", "rgba", "group", "mesh"}
TAG = re.compile(r"<([A-Za-z_][\w]*)((?:\s+[\w:]+\s*=\s*\"[^\"]*\")*)\s*/?>")
ATTR = re.compile(r"([\w:]+)\s*=\s*\"([^\"]*)\"")


def canon(tag, attrs):
    """tag + sorted attributes with whitespace-normalised values; None for rendering-only elements"""
    if tag in VISUAL_TAGS:
        return None
    kept = {k: " ".join(v.split()) for k, v in attrs.items() if k not in VISUAL_ATTRS}
    if tag == "geom" and (kept.get("class") == "visual" or (kept.get("contype") == "0" and kept.get("conaffinity") == "0")):
        return None  # a geom that collides with nothing and carries no mass specification of its own is drawn, not simulated
    if not kept and tag in ("mujoco", "default", "worldbody", "body", "actuator", "sensor", "tendon", "contact", "keyframe", "custom", "option"):
        return None  # bare containers
    return tag + "|" + "|".join(f"{k}={kept[k]}" for k in sorted(kept))


def digest(s):
    return hashlib.sha1(s.encode()).hexdigest()[:16]


def patch_elements(path):
    """canonical elements on the `+` and context lines of a unified diff"""
    text = []
    for line in open(path):
        if line.startswith(("diff ", "--- ", "+++ ", "@@", "index ")):
            text.append("\n")
            continue
        if line.startswith(("+", " ")):
            text.append(line[1:])
    text = re.sub(r"<!--.*?-->", "", "".join(text), flags=re.S)
    out = []
    for m in TAG.finditer(text):
        c = canon(m.group(1), dict(ATTR.findall(m.group(2))))
        if c:
            out.append(c)
    return out


def file_elements(path):
    out = set()
    for el in ET.parse(path).getroot().iter():
        c = canon(el.tag, dict(el.attrib))
        if c:
            out.add(c)
    root = ET.parse(path).getroot()
    c = canon(root.tag, dict(root.attrib))
    if c:
        out.add(c)
    return out


def build():
    fixture = {}
    for name, (patch, _mine) in MODELS.items():
        els = patch_elements(os.path.join(REF, patch))
        fixture[name] = {"patch": "mjpc/tasks/" + patch, "elements": sorted({digest(e) for e in els}), "count": len(set(els))}
    return fixture


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    fixture = build()
    if "--check" in sys.argv:
        for name, (patch, mine) in MODELS.items():
            have = file_elements(os.path.join(ROOT, mine))
            missing = [e for e in set(patch_elements(os.path.join(REF, patch))) if e not in have]
            print(name, "elements", fixture[name]["count"], "missing", len(missing))
            for e in sorted(missing):
                print("   ", e)
        return
    with open(os.path.join(ROOT, "tests", "golden", "model_provenance.json"), "w") as f:
        json.dump(fixture, f, indent=1)
    print({k: v["count"] for k, v in fixture.items()})


if __name__ == "__main__":
    main()
