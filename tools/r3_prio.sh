#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prio; rm -rf $O; mkdir -p $O
go() { echo "=== env: $1 | pre: $2" >> $O/log.txt; env $1 ITERS=16 timeout 300 python $R/tools/r3_single_ctx.py $2 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt; }
go "MAINSTREAM=prio" "c1 c3conc c4 c5conc kron gc"
go "MAINSTREAM=prio" "none"
go "MAINSTREAM=prio" "c1 streams4k"
go "X=0" "c1 c3conc c4 c5conc kron gc"
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -8
