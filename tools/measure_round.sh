#!/bin/bash
# One GPU-box call that collects everything profiles/r02_* is made of: the PMC summary of the current build first (bench.py
# picks it up as roofline.traffic), the default bench line with its extras, and the kernel-trace summary of the same command.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
bash $R/tools/pmc_rollout.sh 16384 64 > $O/pmc.log 2>&1
cp $R/gpurun_out/pmc/r02_pmc_quadrupedflat_fp64.json $R/profiles/ 2>/dev/null   # (this copy lives on the box only; merged back via gpurun_out)
cp $R/gpurun_out/pmc/r02_pmc_quadrupedflat_fp64.json $O/
cd $R
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 3000 $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-extra --no-cpu-baseline > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -8 $O/bench_kernel_stats.csv
tail -3 $O/pmc.log
