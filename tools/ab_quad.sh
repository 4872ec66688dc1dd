#!/bin/bash
# same-box A/B of alternative builds of libmjpcx.so on the north-star line: tools/ab_quad.sh <other.so> [<other2.so> ...]
# (the in-tree library is re-measured between the variants; 8 timed plan steps after 6 warm-up ones, i.e. the planner's gait, not the standing start)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps ${AB_STEPS:-8} --warmup ${AB_WARMUP:-6} --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['roofline']['kernel_ms'],2), d['roofline']['handed_on_last_step']['candidates'])"; }
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
run main
for so in "$@"; do cp $so mujoco_mpc_amd/libmjpcx.so; run $so; cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; run main; done
