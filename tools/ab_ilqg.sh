#!/bin/bash
# same-box A/B of alternative builds of libmjpcx.so on configs[4] (one iLQG iteration on the Quadruped): tools/ab_ilqg.sh <other.so> ...
cd $GRAFT_REPO_ROOT
run() { python -c "
import bench
e = bench.run_ilqg(0, iterations=8, warmup=2, cpu=False)
print('$1', round(e['value'], 2), 'ms per iteration')" 2>/dev/null | tail -1; }
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
run main
for so in "$@"; do cp $so mujoco_mpc_amd/libmjpcx.so; run $so; cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; run main; done
