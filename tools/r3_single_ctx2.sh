#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/singlectx2; rm -rf $O; mkdir -p $O
echo "=== bench.py --extras-only (stderr)" >> $O/log.txt
timeout 600 python $R/bench.py --extras-only > $O/extras.json 2> $O/extras.err; echo "rc=$?" >> $O/log.txt
grep -i "C2 single" $O/extras.err >> $O/log.txt
echo "=== ctx tool, no history, 100 iterations" >> $O/log.txt
ITERS=100 timeout 300 python $R/tools/r3_single_ctx.py none >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt
cat $O/log.txt
