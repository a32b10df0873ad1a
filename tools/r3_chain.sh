#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/chain; rm -rf $O; mkdir -p $O
echo "=== bitwise: chain stream vs caller's stream" >> $O/log.txt
timeout 300 python $R/tools/r3_chain_check.py 16384 6 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt
timeout 300 python $R/tools/r3_chain_check.py 6200 6 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt
go() { echo "=== env: $1 | pre: $2" >> $O/log.txt; env $1 STAGES=1 ITERS=20 timeout 300 python $R/tools/r3_single_ctx.py $2 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt; }
go "X=0" "none"
go "GPIMHIP_NO_CHAIN_STREAM=1" "none"
go "X=0" "c1 streams4"
go "X=0" "c1 c3conc c4 c5conc kron gc"
go "GPIMHIP_NO_CHAIN_STREAM=1" "c1 c3conc c4 c5conc kron gc"
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -8
