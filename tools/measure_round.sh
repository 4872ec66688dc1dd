#!/bin/bash
# One GPU-box call that collects what profiles/${ROUND}_* is made of; the sections run in this order and are picked by WHAT (default: all):
#   tests   the GPU test suite
#   pmc     counters + kernel trace of the launches bench.py times, north star and Humanoid (tools/pmc_bench.sh); bench.py picks the
#           summaries up from profiles/ as roofline.traffic / .valu of its line
#   stamps  phase stamps of wavefront 0 of the quad kernel over the first plan steps of the bench (MJPCX_QUAD_STAMPS), and of the limb kernel on
#           configs[3] (MJPCX_LIMB_STAMPS)
#   fuzz    the random-state parity sweeps (tools/fuzz_quad.py, fuzz_humanoid.py through both Humanoid kernels, fuzz_ilqg.py)
#   bench   the default bench line with its extras
#   trace   rocprofv3 --kernel-trace --stats of the bench command
#   ilqg    kernel trace of the iLQG iteration
# Files land in gpurun_out/$ROUND/ (merged back by gpurun); copy them to profiles/ and commit.
export ROUND=${ROUND:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$ROUND; mkdir -p $O
WHAT=${WHAT:-tests pmc stamps bench trace ilqg fuzz}
has() { [[ " $WHAT " == *" $1 "* ]]; }
cd $R
if has tests; then timeout 900 python -m pytest tests -m gpu -q -x > $O/gputests.log 2>&1; tail -3 $O/gputests.log; fi
if has pmc; then
  bash $R/tools/pmc_bench.sh QuadrupedFlat 64 > $O/pmc_quadrupedflat.log 2>&1; tail -3 $O/pmc_quadrupedflat.log
  bash $R/tools/pmc_bench.sh HumanoidTrack 32 10 2 > $O/pmc_humanoidtrack.log 2>&1; tail -3 $O/pmc_humanoidtrack.log
  for f in $R/gpurun_out/pmc_*/${ROUND}_pmc_*.json; do [ -f $f ] && cp $f $R/profiles/ && cp $f $O/; done   # (the profiles/ copy lives on the box only; $O is merged back)
  for t in quadrupedflat humanoidtrack; do find $R/gpurun_out/pmc_$t/p0 -name "*kernel_trace.csv" -exec cp {} $O/pmc_${t}_kernel_trace.csv \; ; done
fi
cd $R
if has stamps; then MJPCX_QUAD_STAMPS=1 timeout 300 python bench.py --steps 10 --warmup 0 --no-extra --no-cpu-baseline > $O/stamps_line.json 2> $O/quad_stamps.log; grep -c "cycles of wavefront" $O/quad_stamps.log; fi
if has stamps; then MJPCX_LIMB_STAMPS=1 MJPCX_QUAD_STATS=1 timeout 300 python bench.py --task HumanoidTrack --precision 32 --steps 6 --warmup 1 --no-extra --no-cpu-baseline > $O/limb_stamps_line.json 2> $O/limb_stamps.log; grep -c "cycles of wavefront" $O/limb_stamps.log; fi
if has bench; then timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 1500 $O/bench_line.json; fi
cd /tmp && export TMPDIR=/tmp
if has trace; then
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --no-extra --no-cpu-baseline > $O/trace.log 2>&1
  find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
  head -6 $O/bench_kernel_stats.csv
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_h -o bench -- python $R/bench.py --task HumanoidTrack --precision 32 --steps 10 --no-extra --no-cpu-baseline > $O/trace_h.log 2>&1
  find $O/trace_h -name "*kernel_stats.csv" -exec cp {} $O/humanoid_kernel_stats.csv \;
  head -4 $O/humanoid_kernel_stats.csv
  rm -rf $O/trace $O/trace_h
fi
if has ilqg; then bash $R/tools/measure_ilqg.sh > $O/ilqg.log 2>&1; tail -5 $O/ilqg.log; fi
cd $R
if has fuzz; then
  for seed in 1 2 3; do timeout 900 python tools/fuzz_quad.py 300 $seed >> $O/fuzz_quad.log 2>&1; done; tail -2 $O/fuzz_quad.log
  timeout 900 python tools/fuzz_humanoid.py 60 1 64 limb > $O/fuzz_humanoid.log 2>&1; timeout 900 python tools/fuzz_humanoid.py 60 1 64 tree >> $O/fuzz_humanoid.log 2>&1
  timeout 900 python tools/fuzz_humanoid.py 60 2 32 limb >> $O/fuzz_humanoid.log 2>&1; grep "cases x 8" $O/fuzz_humanoid.log
  timeout 900 python tools/fuzz_ilqg.py 40 1 > $O/fuzz_ilqg.log 2>&1; tail -1 $O/fuzz_ilqg.log
fi
true
