#!/bin/bash
# Counter passes over THE LAUNCHES bench.py TIMES (separate --pmc runs, as MI355X_MICROARCH.md prescribes): every pass runs the bench
# command of a task itself -- the C++ planner's warm-up and timed plan steps, so the rollout kernel sees the batches of the headline run,
# launch for launch (the planner is deterministic: same seed, same nominal sequence) -- plus one --kernel-trace pass for the launches'
# durations. Run on the GPU box:   bash tools/pmc_bench.sh [task] [precision] [steps] [warmup]
# writes gpurun_out/pmc_<task>/<pass>/ and the per-build summary gpurun_out/pmc_<task>/${ROUND}_pmc_<task>_fp<prec>.json (tools/pmc_summary.py),
# which bench.py reads from profiles/ (roofline.traffic / .valu) when its source hash matches.
TASK=${1:-QuadrupedFlat}; PREC=${2:-64}; STEPS=${3:-20}; WARM=${4:-2}; ROUND=${ROUND:-r06}
LOW=$(echo $TASK | tr '[:upper:]' '[:lower:]')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$LOW; rm -rf $O; mkdir -p $O
CMD="python $R/bench.py --task $TASK --precision $PREC --steps $STEPS --warmup $WARM --no-extra --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p0 -o p -- $CMD > $O/p0.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -o p -- $CMD > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p -- $CMD > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p -- $CMD > $O/p4.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p5 -o p -- $CMD > $O/p5.log 2>&1
# what the kernel EXECUTES in floating point: instruction counts by class of the working precision, the gfx950 FLOPS
# counter, and thread-level VALU activity (lanes, not wavefronts)
F=F$PREC
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_$F SQ_INSTS_VALU_ADD_$F SQ_INSTS_VALU_MUL_$F SQ_INSTS_VALU_TRANS_$F SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $O/p6 -o p -- $CMD > $O/p6.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_FLOPS_FP$PREC SQ_INSTS_VALU_FLOPS_FP${PREC}_TRANS SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_ACTIVE_INST_VALU2 --output-format csv -d $O/p7 -o p -- $CMD > $O/p7.log 2>&1
python $R/tools/pmc_summary.py $O $TASK $PREC $STEPS $WARM $O/${ROUND}_pmc_${LOW}_fp$PREC.json | head -40
tail -2 $O/p1.log
