#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3_exp9; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tests/tools/prof_fit.py 16384 3 0 Matern52 > $O/log.txt 2>&1
grep -v amdgpu $O/log.txt
