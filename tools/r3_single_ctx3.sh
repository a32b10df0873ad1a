#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/singlectx3; rm -rf $O; mkdir -p $O
for pre in "c1" "c4" "kron" "c1 c3conc" "c1 c3conc c4 c5conc kron gc"; do
  echo "=== pre: $pre" >> $O/log.txt
  ITERS=30 timeout 300 python $R/tools/r3_single_ctx.py $pre >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt
done
cat $O/log.txt; tail -5 $O/err.txt
