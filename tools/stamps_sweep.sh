cd $GRAFT_REPO_ROOT
for s in 2 15 40 70 95; do
  MJPCX_STAMPS=$s timeout 120 python tools/profile_rollout.py --task QuadrupedFlat -n 16384 --horizon 100 --launches 1 --interp 0 --std 0.04 2>&1 | grep -v "^$" | tail -6
done
