#!/bin/bash
# kernel time of the north-star rollout against the batch size: with 2048 resident wavefronts, N = 2048 is one rollout per wavefront
# (time = the slowest rollout), larger N average more rollouts per wavefront (static stride): T/N falling with N measures the tail imbalance
cd $GRAFT_REPO_ROOT
for n in 2048 4096 16384 65536; do timeout 200 python tools/profile_rollout.py --task QuadrupedFlat -n $n --horizon 100 --launches 2 --interp 0 --std 0.04 2>&1 | grep "launches" | sed "s/^/N=$n /" | cut -c1-60,200-400; done
