#!/bin/bash
# Counter passes of a task's rollout kernel (separate --pmc runs, as MI355X_MICROARCH.md prescribes); run on the GPU box:
#   bash tools/pmc_rollout.sh [task] [candidates] [horizon] [precision] [interp] [std]
# writes gpurun_out/pmc_<task>/<pass>/ and the per-build summary gpurun_out/pmc_<task>/r04_pmc_<task>_fp<prec>.json (tools/derive_pmc.py)
TASK=${1:-QuadrupedFlat}; N=${2:-16384}; H=${3:-100}; PREC=${4:-64}; INTERP=${5:-0}; STD=${6:-0.04}
LOW=$(echo $TASK | tr '[:upper:]' '[:lower:]')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$LOW; rm -rf $O; mkdir -p $O
CMD="python $R/tools/profile_rollout.py --task $TASK -n $N --horizon $H --launches 2 --interp $INTERP --std $STD --precision $PREC"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d $O/p2 -o p -- $CMD > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p -- $CMD > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p -- $CMD > $O/p4.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p5 -o p -- $CMD > $O/p5.log 2>&1
python $R/tools/summarize_pmc.py $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 --kernel=rollout_ > $O/summary.json
python $R/tools/derive_pmc.py $O/summary.json $TASK $N $H $PREC $O/r04_pmc_${LOW}_fp$PREC.json | head -30
tail -2 $O/p1.log
