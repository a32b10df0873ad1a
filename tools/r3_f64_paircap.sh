#!/bin/bash
# fp64, pair mode: does hosting a few bulk tiles per step launch pay after all?  (potrf stage of a fit at N = 16384 / 20480)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/paircap; rm -rf $O; mkdir -p $O
run() { echo "--- N=$1 $2" >> $O/log.txt; env $2 PROF_STAGES=1 timeout 300 python $R/tests/tools/prof_fit.py $1 8 0 Matern52 2>> $O/err.txt | grep "potrf" | tail -1 >> $O/log.txt; }
for n in 16384; do
  run $n "X=0"
  for cap in 4 8 16 32; do run $n "GPIMHIP_FILL_CAP=$cap"; done
  run $n "GPIMHIP_FILL_CAP=8 GPIMHIP_FILL_QUAD_MAX=256"
  run $n "GPIMHIP_FILL_CAP=16 GPIMHIP_FILL_QUAD_MAX=256"
  run $n "GPIMHIP_FILL_QUAD_MAX=256"
  run $n "GPIMHIP_FILL_QUAD_MAX=64"
  run $n "GPIMHIP_FILL_HALF_MAX=256"
done
cat $O/log.txt
