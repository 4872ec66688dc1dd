cd $GRAFT_REPO_ROOT
run() {
  a=$(python bench.py --task Cartpole --steps 30 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['roofline']['kernel_ms'],4))")
  b=$(python tools/quad_n_sweep.py 2>/dev/null | grep -E "tree N =  (2048|4096)" | tr '\n' ' ')
  c=$(python -c "
import bench
e = bench.run_ilqg(0, iterations=6, warmup=2)
print(round(e['value'], 2), round(e['backward_pass']['kernel_ms'],3))" 2>/dev/null | tail -1)
  d=$(python bench.py --task HumanoidTrack --candidates 2048 --horizon 64 --precision 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e3,1))")
  echo "$1: Cartpole $a | $b | iLQG $c | Humanoid fp64 $d k"
}
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
run main
cp variants/lib_M.so mujoco_mpc_amd/libmjpcx.so; run M; cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; run main
