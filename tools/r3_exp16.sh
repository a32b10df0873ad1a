#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_e2e.py tests/test_gpu_highprec.py tests/test_gpu_fullsize.py tests/test_gpu_single.py -q -m gpu 2>&1 | tail -5
{
for n in 1207 4206; do PROF_STAGES=1 python tests/tools/prof_fit.py $n 30 0 RBF | grep -v workspace; done
python tests/tools/bench_bo.py 2>/dev/null | tail -2
} 2>&1 | grep -v amdgpu
