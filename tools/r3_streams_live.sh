#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/streamslive; rm -rf $O; mkdir -p $O
go() { echo "=== pre: $1" >> $O/log.txt; PRECS=double STAGES=1 ITERS=16 timeout 300 python $R/tools/r3_single_ctx.py $1 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt; }
go "c1 streams4k"
go "c1 streams4d"
go "c1 streams4d streams4d streams4d"
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -8
