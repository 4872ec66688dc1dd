#!/bin/bash
# PMC passes over tools/quad_prof.py (run on the GPU box): separate --pmc runs as the MI355X guide prescribes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qpmc; mkdir -p $O
CMD="python $R/tools/quad_prof.py 2"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -o p -- $CMD > $O/p2.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p -- $CMD > $O/p3.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p -- $CMD > $O/p4.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o p -- $CMD > $O/kt.log 2>&1
ls -R $O | head -40
