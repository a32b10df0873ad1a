#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/singlectx6; rm -rf $O; mkdir -p $O
go() { echo "=== env: $1 | pre: $2" >> $O/log.txt; env $1 STAGES=1 ITERS=12 PRECS=double MIDN=1 timeout 300 python $R/tools/r3_single_ctx.py $2 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt; }
go "X=0" "none"
go "X=0" "c1 streams4"
go "MAINSTREAM=cumask" "none"
go "MAINSTREAM=cumask" "c1 streams4"
go "MAINSTREAM=prio" "c1 streams4"
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -8
