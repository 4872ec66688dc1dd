mkdir -p gpurun_out/c1
{ python -c "import mujoco; print('mujoco', mujoco.__version__)"; echo "rc=$?"; pip download mujoco==3.1.6 -d /tmp/mj 2>&1 | tail -3; echo "rc=$?"; find / -xdev \( -name 'libmujoco*' -o -name 'mujoco*.whl' -o -name 'mujoco' -type d \) 2>/dev/null | grep -v "^/root/repo\|graft" | head; echo "find done"; pip list 2>/dev/null | grep -i -E "mujoco|dm_control|gym|brax|mjx" ; echo "pip list done"; } > gpurun_out/c1/mujoco_probe.log 2>&1
rocm-smi --showclocks --showpower --showperflevel > gpurun_out/c1/smi_idle.log 2>&1
which amd-smi rocm-smi >> gpurun_out/c1/smi_idle.log 2>&1
python bench.py --steps 10 --warmup 2 --no-extra > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err
MJPCX_QUAD_STAMPS=1 timeout 300 python bench.py --no-extra --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/c1/stamps.log 2>&1
nproc >> gpurun_out/c1/smi_idle.log; cat /sys/fs/cgroup/cpu.max >> gpurun_out/c1/smi_idle.log 2>&1
