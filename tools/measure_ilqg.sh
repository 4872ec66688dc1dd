#!/bin/bash
# kernel trace of one Quadruped iLQG planning run (configs[4]) -> gpurun_out/$ROUND/ilqg_kernel_stats.csv (copy to profiles/${ROUND}_ilqg_kernel_stats.csv)
ROUND=${ROUND:-r06}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ilqg_trace -o ilqg -- python -c "
import sys; sys.path.insert(0, '$R')
import bench
print(bench.run_ilqg(0, iterations=6, warmup=2))
" > $O/ilqg_trace.log 2>&1
find $O/ilqg_trace -name "*kernel_stats.csv" -exec cp {} $O/ilqg_kernel_stats.csv \;
head -12 $O/ilqg_kernel_stats.csv | cut -c1-200
tail -2 $O/ilqg_trace.log | cut -c1-1200
