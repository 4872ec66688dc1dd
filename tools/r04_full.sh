#!/bin/bash
# the GPU suite, the default bench line and the sysfs clock files (run on the GPU box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/full
ls /sys/class/drm/ > gpurun_out/full/sysfs.log 2>&1; for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h; ls $h | tr '\n' ' '; echo; cat $h/freq1_input $h/power1_average $h/power1_input 2>&1; done >> gpurun_out/full/sysfs.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/full/tests.log
python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err
tail -3 gpurun_out/full/tests.log
