#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/graphlarge; rm -rf $O; mkdir -p $O
echo "=== bitwise: captured vs eager, N=16384" >> $O/log.txt
timeout 300 python $R/tools/r3_graph_large_check.py 16384 12 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt
echo "=== bitwise: captured vs eager, N=6200 (just inside the large regime)" >> $O/log.txt
timeout 300 python $R/tools/r3_graph_large_check.py 6200 12 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt
go() { echo "=== env: $1 | pre: $2" >> $O/log.txt; env $1 ITERS=20 timeout 300 python $R/tools/r3_single_ctx.py $2 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt; }
go "X=0" "none"
go "GPIMHIP_NO_GRAPH_LARGE=1" "none"
go "X=0" "c1 streams4"
go "GPIMHIP_NO_GRAPH_LARGE=1" "c1 streams4"
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -8
