cd $GRAFT_REPO_ROOT
run() { python bench.py --task HumanoidTrack --candidates 8192 --horizon 64 --precision 32 --steps 6 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']/1e3,1), 'k rollouts/s', round(d['roofline']['kernel_ms'],2), 'ms')"; }
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
run main
MJPCX_TREE_WAVES=11 run "main 11 waves"
cp variants/lib_640.so mujoco_mpc_amd/libmjpcx.so; MJPCX_TREE_WAVES=10 run "640 threads, 10 wavefronts"
cp variants/lib_512.so mujoco_mpc_amd/libmjpcx.so; MJPCX_TREE_WAVES=8 run "512 threads, 8 wavefronts"
cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; run main
