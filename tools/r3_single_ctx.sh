#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/singlectx; rm -rf $O; mkdir -p $O
for pre in "none" "c3serial" "c3conc" "c3conc gc" "c5conc" "c5conc gc"; do
  echo "=== pre: $pre" >> $O/log.txt
  timeout 300 python $R/tools/r3_single_ctx.py $pre >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt
done
cat $O/log.txt
