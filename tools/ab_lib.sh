#!/bin/bash
# same-box A/B of alternative builds of libmjpcx.so on three lines: the north star (quad kernel), configs[3] (fp32 tree kernel of the
# Humanoid) and configs[4] (iLQG iteration: feedback rollouts, sweep, backward pass): tools/ab_lib.sh <other.so> [<other2.so> ...]
cd $GRAFT_REPO_ROOT
run() {
  a=$(python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],2))")
  b=$(python bench.py --task HumanoidTrack --candidates 8192 --horizon 64 --precision 32 --steps 5 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e3,1))")
  c=$(python -c "
import bench
e = bench.run_ilqg(0, iterations=6, warmup=2, cpu=False) if 'cpu' in bench.run_ilqg.__code__.co_varnames else bench.run_ilqg(0, iterations=6, warmup=2)
print(round(e['value'], 2))" 2>/dev/null | tail -1)
  echo "$1: north-star kernel $a ms | Humanoid fp32 $b k rollouts/s | iLQG iteration $c ms"
}
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
run main
for so in "$@"; do cp $so mujoco_mpc_amd/libmjpcx.so; run $so; cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; run main; done
