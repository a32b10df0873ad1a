#!/bin/bash
# round 6, session 8: new tests (RCCL world 1, fault injection, C4 instance, campaign, fused finalize), grouped K^-1 pass of f1
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s8; rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_dist2.py -x -q) > $O/tests_dist.log 2>&1; tail -5 $O/tests_dist.log
(time timeout 900 python -m pytest tests/test_gpu_e2e.py "tests/test_gpu_regimes.py::test_fused_finalize_equals_two_launches" -x -q) > $O/tests_e2e.log 2>&1; tail -5 $O/tests_e2e.log
for g in 1 4 8 4 1; do
  echo "== KINV_GROUP=$g" >> $O/f1.log
  GPIM_DIST_KINV_GROUP=$g timeout 300 python tools/dist_fit_time.py 65536 2 2>&1 | grep -v amdgpu >> $O/f1.log
done
cat $O/f1.log
for g in 1 4; do
  echo "== rank share, KINV_GROUP=$g" >> $O/share.log
  GPIM_DIST_KINV_GROUP=$g timeout 600 python tools/r5_dist_rank_share.py 65536 1 8 2>&1 | grep -v amdgpu >> $O/share.log
done
cat $O/share.log
