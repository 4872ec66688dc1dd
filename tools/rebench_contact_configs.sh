cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rebench
python bench.py --task QuadrupedFlat --planner cross_entropy --candidates 16384 --horizon 100 --steps 3 --warmup 1 > gpurun_out/rebench/quadruped_ce.json 2>gpurun_out/rebench/err1
python bench.py --task QuadrupedFlat --planner sampling --candidates 16384 --horizon 100 --steps 3 --warmup 1 > gpurun_out/rebench/quadruped_ps.json 2>gpurun_out/rebench/err2
python bench.py --task QuadrupedFlat --planner cross_entropy --candidates 16384 --horizon 100 --steps 3 --warmup 1 --precision 32 > gpurun_out/rebench/quadruped_ce_fp32.json 2>gpurun_out/rebench/err3
python bench.py --task QuadrupedFlat --planner sampling --candidates 16384 --horizon 100 --steps 3 --warmup 1 --precision 32 > gpurun_out/rebench/quadruped_ps_fp32.json 2>gpurun_out/rebench/err4
python bench.py --task HumanoidTrack --candidates 8192 --horizon 64 --steps 3 --warmup 1 > gpurun_out/rebench/humanoid_ps.json 2>gpurun_out/rebench/err5
python bench.py --task HumanoidTrack --candidates 8192 --horizon 64 --steps 3 --warmup 1 --precision 32 > gpurun_out/rebench/humanoid_ps_fp32.json 2>gpurun_out/rebench/err6
for f in gpurun_out/rebench/*.json; do echo $f; tail -1 $f | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"; done
