#!/bin/bash
# Stream-population effect: how does the slow-down depend on WHICH hardware queue the engine's side stream lands on?
# c1, then k extra streams (k = 0..8), then the N = 16384 fit driven from the caller's stream; per k the hardware queue ids
# of the main stream and of the side stream (the mat-vecs beside the K^-1 product) and the time per iteration.
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_queue_sweep; rm -rf $O; mkdir -p $O
for k in 0 1 2 3 4 5 6 7 8; do
  GPIMHIP_NO_CHAIN_STREAM=1 PRECS=double ITERS=3 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tools/r3_single_ctx.py c1 streams$k > $O/log.txt 2>&1
  f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
  echo "k=$k $(grep 'double:' $O/log.txt) | $(python $GRAFT_REPO_ROOT/tools/r4_queue_gaps.py $f | grep -E '^training iteration|^queue ' | tr '\n' ' ')" >> $O/sweep.txt
  rm -rf $O/kt
done
cat $O/sweep.txt
