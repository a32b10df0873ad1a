#!/bin/bash
cd $GRAFT_REPO_ROOT
{
for tb in 0 48 72 96; do echo "== TAIL_BLOCKS=$tb"; GPIMHIP_TAIL_BLOCKS=$tb python tools/potrf_run.py 10240 12288 16384 20480; done
echo "== in-fit, default"; PROF_STAGES=1 python tests/tools/prof_fit.py 16384 4 0 Matern52 | grep -v workspace
} 2>&1 | grep -v "amdgpu"
