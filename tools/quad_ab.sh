#!/bin/bash
# one GPU call while tuning the quad kernel: the north-star line (main-kernel ms / all rollout kernels ms / handed on), the cycle stamps of
# wavefront 0 on the same workload, and the bring-up check (parity of 64 candidates against the oracle, zero-nominal batch time)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --no-extra --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/b1.log 2>&1
tail -1 gpurun_out/b1.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms'], r['all_rollout_kernels_ms'], r['handed_on_last_step'])"
MJPCX_QUAD_STAMPS=1 timeout 300 python bench.py --no-extra --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep "cycles of wavefront" | tail -2
MJPCX_QUAD_STATS=1 timeout 200 python tools/quad_check.py 2>&1 | grep -E "std|N =|tree" | tail -5
