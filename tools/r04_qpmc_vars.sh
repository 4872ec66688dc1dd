#!/bin/bash
# r04_qpmc_vars.sh <variant.so>...: the quick counter passes (tools/r04_qpmc.sh) on main and on each variant library
cd $GRAFT_REPO_ROOT
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
for so in "$@"; do n=$(basename $so .so); cp $so mujoco_mpc_amd/libmjpcx.so; echo "== $n"; bash tools/r04_qpmc.sh $n 2>&1 | grep -E "VALU  |VMEM|WAVE_CYCLES|frac|LDS  "; done
cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so
