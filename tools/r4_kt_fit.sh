#!/bin/bash
# kernel trace of a short fit at N = $1 (T = $2 iterations, kernel $3); per-kernel sums of the LAST iteration
cd /tmp; export TMPDIR=/tmp
N=$1; T=$2; K=${3:-RBF}
O=$GRAFT_REPO_ROOT/gpurun_out/r4_ktfit_${TAG:-x}; rm -rf $O; mkdir -p $O
GPIMHIP_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tests/tools/prof_fit.py $N $T 0 $K > $O/log.txt 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/r4_kt_iter.py $f > $O/iter.txt 2>&1
rm -rf $O/kt
tail -4 $O/log.txt; head -60 $O/iter.txt
