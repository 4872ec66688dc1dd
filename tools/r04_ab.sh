#!/bin/bash
# same-box A/B on the north-star line: the tree's libmjpcx.so ("main") against variants/*.so given as arguments; then the quad GPU tests on main
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d.get('gpu_clock') or {}; print('$1', round(d['value']), round(d['roofline']['kernel_ms'],2), round(d['ms_per_step'],2), 'sclk', round(c.get('sclk_mhz_mean',0)), 'W', round(c.get('power_w_mean',0)), 'mclk', round(c.get('mclk_mhz_mean',0)), 'fclk', c.get('fclk_mhz'), 'socclk', c.get('socclk_mhz'))"; }
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
{
run main
for so in "$@"; do cp $so mujoco_mpc_amd/libmjpcx.so; run $so; cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so; run main; done
} | tee gpurun_out/ab/ab.log
if [ -n "$AB_TESTS" ]; then timeout 600 python -m pytest $AB_TESTS -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/ab/tests.log; fi
