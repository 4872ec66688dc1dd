#!/bin/bash
# same-box A/B on configs[3] (Humanoid fp32, 8192 x 64): the tree's libmjpcx.so against the variant libraries given
cd $GRAFT_REPO_ROOT
run() { python bench.py --task HumanoidTrack --candidates 8192 --horizon 64 --precision 32 --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['roofline']['kernel_ms'],2))"; }
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
run main
for so in "$@"; do cp $so mujoco_mpc_amd/libmjpcx.so; run $so; cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; run main; done
