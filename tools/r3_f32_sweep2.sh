#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/f32sweep2; rm -rf $O; mkdir -p $O
run() { echo "--- N=$1 $2" >> $O/log.txt; env $2 PROF_PRECISION=single PROF_STAGES=1 timeout 300 python $R/tests/tools/prof_fit.py $1 12 0 Matern52 2>> $O/err.txt | grep "potrf" | tail -1 >> $O/log.txt; }
for n in 6400 8192 10240 12288 16384 20480; do
  for cap in 64 128 256 512 100000; do run $n "GPIMHIP_PAIR=0 GPIMHIP_FILL_CAP=$cap"; done
  run $n "GPIMHIP_PAIR=1 GPIMHIP_FILL_CAP=64"
done
cat $O/log.txt
