#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/singlectx5; rm -rf $O; mkdir -p $O
go() { echo "=== env: $1 | pre: $2" >> $O/log.txt; env $1 STAGES=1 ITERS=12 PRECS=double timeout 300 python $R/tools/r3_single_ctx.py $2 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt; }
go "X=0" "c1 streams4"
go "GPU_MAX_HW_QUEUES=8" "c1 streams4"
go "GPU_MAX_HW_QUEUES=16" "c1 streams4"
go "GPIMHIP_NO_GRAPH=1" "c1 streams4"
go "GPIMHIP_NO_CUMASK=1" "c1 streams4"
go "MAINSTREAM=1" "c1 streams4"
go "MAINSTREAM=1" "none"
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -5
