#!/bin/bash
# Stream-population effect: kernel traces of the N = 16384 fit driven from the CALLER's stream
# (GPIMHIP_NO_CHAIN_STREAM=1) in a clean process and after "c1 streams4"; per case the hardware queue of the
# factorisation's launches and the distribution of the gaps between consecutive launches of the chain.
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_queue; rm -rf $O; mkdir -p $O
for case in fast:none slow:"c1 streams4"; do
  tag=${case%%:*}; pre=${case#*:}
  GPIMHIP_NO_CHAIN_STREAM=1 PRECS=double ITERS=3 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$tag -- python $GRAFT_REPO_ROOT/tools/r3_single_ctx.py $pre > $O/log_$tag.txt 2>&1
  f=$(find $O/kt_$tag -name '*kernel_trace.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/r4_queue_gaps.py $f > $O/gaps_$tag.txt 2>&1
  rm -rf $O/kt_$tag
  grep "double:" $O/log_$tag.txt; cat $O/gaps_$tag.txt
done
