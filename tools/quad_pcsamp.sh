#!/bin/bash
# PC sampling of the quad kernel (tools/quad_prof.py): where the wavefronts' program counters are, by instruction
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pcs; rm -rf $O; mkdir -p $O
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${1:-stochastic} --pc-sampling-unit ${2:-cycles} --pc-sampling-interval ${3:-1048576} --output-format csv -d $O -o p -- python $R/tools/quad_prof.py 1 > $O/log.txt 2>&1
echo rc=$?
tail -5 $O/log.txt
ls -la $O | head
