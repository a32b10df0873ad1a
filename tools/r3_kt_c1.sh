#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_kt_c1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tests/tools/prof_fit.py 4206 12 0 RBF > $O/log.txt 2>&1
