#!/bin/bash
# same-box A/B of alternative builds of libmjpcx.so on configs[3] (Humanoid tracking, 8192 x 64, fp32): tools/ab_humanoid.sh <other.so> ...
cd $GRAFT_REPO_ROOT
run() { python bench.py --task HumanoidTrack --candidates 8192 --horizon 64 --precision 32 --steps 5 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e3,1), 'k rollouts/s', round(d['roofline']['kernel_ms'],2), 'ms')"; }
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
run main
for so in "$@"; do cp $so mujoco_mpc_amd/libmjpcx.so; run $so; cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; run main; done
