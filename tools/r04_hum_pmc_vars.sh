#!/bin/bash
# HBM traffic + counters of configs[3] (Humanoid fp32) for the tree's library and for each variant library given (same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/humvars
cp mujoco_mpc_amd/libmjpcx.so /tmp/main.so
for so in main "$@"; do
  [ $so != main ] && cp $so mujoco_mpc_amd/libmjpcx.so
  bash tools/pmc_rollout.sh HumanoidTrack 8192 64 32 2 0.1 > gpurun_out/humvars/pmc.log 2>&1
  python - "$so" <<'PY'
import json,glob,sys
f=sorted(glob.glob('gpurun_out/pmc_*/r04_pmc_humanoidtrack_fp32.json'))[-1]
d=json.load(open(f)); v=d['valu']
print(sys.argv[1], 'traffic GB', round(d['hbm_bytes_per_launch']/1e9,2), 'VMEM_RD', round(v['per_wavefront_step']['VMEM_RD']), 'VMEM_WR', round(v['per_wavefront_step']['VMEM_WR']), 'VALU', round(v['per_wavefront_step']['VALU']), 'LDS', round(v['per_wavefront_step']['LDS']), 'wait', round(v['wait_any_frac'],3), 'valu', round(v['active_valu_frac'],3))
PY
  cp /tmp/main.so mujoco_mpc_amd/libmjpcx.so
done
