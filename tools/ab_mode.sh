#!/bin/bash
# same-box A/B of a run-time kernel mode of the registered-model kernel: tools/ab_mode.sh <MJPCX_TREE_MODE value>
cd $GRAFT_REPO_ROOT
run() { MJPCX_TREE_MODE=$1 timeout 120 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $1', round(d['value']), round(d['roofline']['kernel_ms'],2))"; }
run 0; run $1; run 0; run $1
