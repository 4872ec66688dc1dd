#!/bin/bash
# everything profiles/r04_* holds that depends on the wavefront-per-candidate kernels, in one GPU-box call: the round's PMC + bench + trace
# (measure_round.sh), the iLQG trace, the three random-state sweeps of those kernels, the latency probe and the strong-scaling emulation
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
bash $R/tools/measure_round.sh > $O/measure_round.log 2>&1
bash $R/tools/measure_ilqg.sh > $O/measure_ilqg.log 2>&1
cd $R
{ timeout 300 python tools/fuzz_quad.py 150 1 tree; } > $O/fuzz_tree.log 2>&1
timeout 300 python tools/fuzz_humanoid.py 90 1 > $O/fuzz_humanoid.log 2>&1
timeout 300 python tools/fuzz_ilqg.py 40 1 > $O/fuzz_ilqg.log 2>&1
timeout 300 python tools/latency_probe.py > $O/latency_probe.log 2>&1
timeout 300 python bench.py --scaling strong --emulate-world 8 --no-cpu-baseline > $O/strong_scaling_emulation_8.json 2> $O/strong.err
tail -2 $O/fuzz_tree.log $O/fuzz_humanoid.log $O/fuzz_ilqg.log | cut -c1-300
tail -c 600 $O/bench_line.json
