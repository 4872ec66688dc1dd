#!/bin/bash
# same-box A/B of two builds of libmjpcx.so on the four contact-model bench lines: tools/ab_libs.sh <other.so> (relative to the repo root)
cd $GRAFT_REPO_ROOT
run() { for t in "QuadrupedFlat --candidates 16384 --horizon 100" "HumanoidTrack --candidates 8192 --horizon 64"; do for p in 64 32; do python bench.py --task $t --planner sampling --steps 3 --warmup 1 --precision $p --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['config']['workload'][:14], d['dtype'], round(d['value']), round(d['roofline']['kernel_ms'],1))"; done; done; }
run new
cp mujoco_mpc_amd/libmjpcx.so /tmp/new.so; cp $1 mujoco_mpc_amd/libmjpcx.so
run other
cp /tmp/new.so mujoco_mpc_amd/libmjpcx.so
run new
