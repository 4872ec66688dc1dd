#!/bin/bash
# instruction-fetch and issue counters of the quad kernel (tools/quad_prof.py: two launches of the north-star batch); separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qpmc2; rm -rf $O; mkdir -p $O
CMD="python $R/tools/quad_prof.py 2"
timeout 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $O/p1 -o p -- $CMD > $O/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/p2 -o p -- $CMD > $O/p2.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p3 -o p -- $CMD > $O/p3.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES --output-format csv -d $O/p4 -o p -- $CMD > $O/p4.log 2>&1
python - <<'PY'
import csv, glob, collections, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/qpmc2"
for f in sorted(glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "rollout_quad" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f.split("/")[-3], {k: v / max(n[k], 1) for k, v in acc.items()})
PY
