#!/bin/bash
# hosting thresholds / pair mode for the FLOAT step schedule (defaults were tuned on fp64)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/f32sweep; rm -rf $O; mkdir -p $O
run() { echo "--- N=$1 $2" >> $O/log.txt; env $2 PROF_PRECISION=single PROF_STAGES=1 timeout 300 python $R/tests/tools/prof_fit.py $1 12 0 Matern52 2>> $O/err.txt | grep "potrf\|ms/iter" | tail -2 >> $O/log.txt; }
for n in 2560 4224; do
  run $n "X=0"
  run $n "GPIMHIP_FILL_QUAD_MAX=64"
  run $n "GPIMHIP_FILL_QUAD_MAX=256"
  run $n "GPIMHIP_FILL_QUAD_MAX=256 GPIMHIP_FILL_HALF_MAX=1024"
  run $n "GPIMHIP_FILL_HALF_MAX=256"
done
for n in 8192 12288 16384; do
  run $n "X=0"
  run $n "GPIMHIP_PAIR=0"
  run $n "GPIMHIP_PAIR=1"
  run $n "GPIMHIP_PAIR=0 GPIMHIP_FILL_CAP=0"
  run $n "GPIMHIP_PAIR=0 GPIMHIP_FILL_CAP=128"
done
cat $O/log.txt
