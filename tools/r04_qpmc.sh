#!/bin/bash
# quick counter passes over the quad kernel (tools/quad_prof.py: north-star batch), separate --pmc runs; prints per wavefront-step figures
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qpmc_${1:-x}; rm -rf $O; mkdir -p $O
CMD="python $R/tools/quad_prof.py 2"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -o p -- $CMD > $O/p2.log 2>&1
if [ -n "$QPMC_HBM" ]; then
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p3 -o p -- $CMD > $O/p3.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p4 -o p -- $CMD > $O/p4.log 2>&1
fi
python $R/tools/summarize_pmc.py $O/p1 $O/p2 $O/p3 $O/p4 --kernel=rollout_quad > $O/summary.json
python - <<PY | tee $O/derived.txt
import json
d=json.load(open("$O/summary.json"))
for k,v in d.items():
    ws=v.get("SQ_WAVES",1024)*100.0
    print(k[-40:])
    for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_INSTS_FLAT","SQ_INSTS_SMEM","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_ACTIVE_INST_VALU","SQ_WAIT_INST_LDS"):
        if c in v: print("  %-22s %12.1f per wavefront-step" % (c, v[c]/ws))
    if "SQ_WAIT_ANY" in v: print("  wait_any_frac %.3f active_valu_frac %.3f" % (v["SQ_WAIT_ANY"]/v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_VALU"]*4/v["SQ_WAVE_CYCLES"]))
    if "FETCH_SIZE" in v: print("  FETCH_SIZE KiB %.0f (x2 on gfx950) WRITE_SIZE KiB %.0f -> GB %.2f" % (v["FETCH_SIZE"], v.get("WRITE_SIZE",0), (2*v["FETCH_SIZE"]+v.get("WRITE_SIZE",0))*1024/1e9))
PY
