#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3_exp3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt_new -- python $GRAFT_REPO_ROOT/tools/potrf_run.py 4224 16384 > $O/new.log 2>&1
cd $GRAFT_REPO_ROOT
for n in 1207 4206; do PROF_STAGES=1 python tests/tools/prof_fit.py $n 40 0 RBF; done 2>&1 | grep -v amdgpu.ids
