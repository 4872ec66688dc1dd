#!/bin/bash
# memory-instruction counts and outstanding-request levels of the quad kernel (average latency = LEVEL / INSTS)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qpmc3; rm -rf $O; mkdir -p $O
CMD="python $R/tools/quad_prof.py 2"
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT --output-format csv -d $O/p2 -o p -- $CMD > $O/p2.log 2>&1
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/qpmc3"
for f in sorted(glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "rollout_quad" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f.split("/")[-2], {k: v / max(n[k], 1) for k, v in acc.items()})
PY
