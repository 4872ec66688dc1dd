#!/usr/bin/env python3
"""Converts the reference's Humanoid-tracking keyframe XML files (mjpc/tasks/humanoid/tracking/keyframes/*.xml, CMU mocap
data: mocap.cs.cmu.edu, created with funding from NSF EIA-0196217) into one compact table,
mujoco_mpc_amd/models/humanoid/tracking/tracking_keyframes.npz, in the <include> order of the reference's task.xml:
    names[nkey], mpos[nkey, 48], qpos[nkey, 28] (NaN rows where a key has none -> qpos0 at load time), qvel[nkey, 27].
Run in the build container (reads /root/reference); the table travels with the repo."""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

ORDER = ["02-02_04", "87-87_01", "88-88_06", "88-88_07", "88-88_08", "88-88_09", "90-90_19", "103-103_08", "108-108_13", "137-137_40"]
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/mjpc/tasks/humanoid/tracking/keyframes"
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names, mpos, qpos, qvel, lengths = [], [], [], [], []
for tag in ORDER:
    root = ET.parse(os.path.join(src, f"CMU-CMU-{tag}_poses.xml")).getroot()
    keys = [k for sec in root.findall("keyframe") for k in sec]
    lengths.append(len(keys))
    for k in keys:
        names.append(k.get("name"))
        mpos.append(np.array(k.get("mpos").split(), float))
        qpos.append(np.array(k.get("qpos").split(), float) if k.get("qpos") else np.full(28, np.nan))
        qvel.append(np.array(k.get("qvel").split(), float) if k.get("qvel") else np.zeros(27))
out = os.path.join(here, "mujoco_mpc_amd", "models", "humanoid", "tracking", "tracking_keyframes.npz")
np.savez_compressed(out, names=np.array(names), mpos=np.array(mpos), qpos=np.array(qpos), qvel=np.array(qvel),
                    lengths=np.array(lengths, np.int32))
print(out, len(names), "keys; motion lengths", lengths)
