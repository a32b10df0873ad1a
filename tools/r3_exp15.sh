#!/bin/bash
cd $GRAFT_REPO_ROOT
{
python tools/potrf_run.py 4224 16384
for n in 4206 16384; do T=30; [ $n -ge 8192 ] && T=4; PROF_STAGES=1 python tests/tools/prof_fit.py $n $T 0 Matern52 | grep -v workspace; done
} 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py -q -m gpu -x 2>&1 | tail -2
