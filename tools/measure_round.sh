#!/bin/bash
# One GPU-box call that collects everything profiles/r04_* is made of: the GPU test suite, the PMC summary of the north-star kernel
# (bench.py picks it up as roofline.traffic / .valu of the line), the default bench line with its extras, the kernel-trace summary of the
# same command, then the PMC summaries of the other two configurations. The files land in gpurun_out/r04/ (merged back by gpurun);
# copy them to profiles/ and commit.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 300 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
[ -z "$SKIP_PMC" ] && bash $R/tools/pmc_rollout.sh QuadrupedFlat 16384 100 64 0 0.04 > $O/pmc_quadrupedflat.log 2>&1   # (SKIP_PMC=1: the committed profiles/r04_pmc_*.json are of this build)
for f in $R/gpurun_out/pmc_*/r04_pmc_*.json; do [ -f $f ] && cp $f $R/profiles/ && cp $f $O/; done   # (the profiles/ copy lives on the box only; $O is merged back)
cd $R
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 1500 $O/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-extra --no-cpu-baseline > $O/trace.log 2>&1
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
head -6 $O/bench_kernel_stats.csv
[ -n "$SKIP_PMC" ] && exit 0
tail -3 $O/pmc_quadrupedflat.log
bash $R/tools/pmc_rollout.sh HumanoidTrack 8192 64 32 2 0.1 > $O/pmc_humanoidtrack.log 2>&1
bash $R/tools/pmc_rollout.sh Cartpole 4096 128 64 2 0.5 > $O/pmc_cartpole.log 2>&1
for f in $R/gpurun_out/pmc_*/r04_pmc_*.json; do cp $f $O/; done
