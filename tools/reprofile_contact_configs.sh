#!/bin/bash
# rocprofv3 kernel-trace summaries of the contact-model bench lines (the commands profiles/README.md lists); run on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/reprof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/quad_ce -o p -- python $R/bench.py --task QuadrupedFlat --planner cross_entropy --candidates 4096 --horizon 100 --steps 2 --warmup 1 --no-cpu-baseline > $O/quad_ce.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hum -o p -- python $R/bench.py --task HumanoidTrack --planner sampling --candidates 8192 --horizon 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/hum.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/hum32 -o p -- python $R/bench.py --task HumanoidTrack --planner sampling --candidates 8192 --horizon 64 --steps 2 --warmup 1 --precision 32 --no-cpu-baseline > $O/hum32.log 2>&1
ls $O/*/
