#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/singlectx4; rm -rf $O; mkdir -p $O
for pre in "none" "c1 c3conc" "c3conc c1" "c1 streams4" "c1 c3groups" "streams4"; do
  echo "=== pre: $pre" >> $O/log.txt
  STAGES=1 ITERS=20 timeout 300 python $R/tools/r3_single_ctx.py $pre >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt
done
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -5
