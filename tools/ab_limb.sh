#!/bin/bash
# same-box A/B of alternative builds of libmjpcx.so on configs[3] (Humanoid tracking, 8192 x 64, fp32) with the limb kernel's phase stamps:
#   tools/ab_limb.sh <variant.so> ...      (variants from tools/build_variant.py --unit limb_kernel; env CPW="0 16" picks candidates per wavefront)
cd $GRAFT_REPO_ROOT
run() {
  for cpw in ${CPW:-0}; do
    MJPCX_QUAD_STATS=1 MJPCX_LIMB_STAMPS=1 MJPCX_LIMB_CPW=$cpw python bench.py --task HumanoidTrack --candidates 8192 --horizon 64 --precision 32 --steps 5 --warmup 1 --no-cpu-baseline --no-extra 2> /tmp/ab.err | tail -1 |
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 cpw $cpw:', round(d['value']/1e3,1), 'k rollouts/s, kernel', round(d['roofline']['kernel_ms'],2), 'ms, all rollout kernels', round(d['roofline']['all_rollout_kernels_ms'],2), 'ms')"
    grep "cycles of wavefront" /tmp/ab.err | tail -1; grep "newton: setup" /tmp/ab.err | tail -1; grep "line-search der" /tmp/ab.err | tail -1; grep "forward: kin" /tmp/ab.err | tail -1; grep "residual: joint" /tmp/ab.err | tail -1; grep "raw 32" /tmp/ab.err | tail -1; grep "handed to" /tmp/ab.err | tail -1
  done
}
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
[ -z "$SKIP_MAIN" ] && run main
for so in "$@"; do cp $so mujoco_mpc_amd/libmjpcx.so; run $so; cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; done
