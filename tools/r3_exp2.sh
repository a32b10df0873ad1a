#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp2; mkdir -p $O
{
echo "== new step schedule"; python tools/potrf_run.py 128 384 1280 4224 8192 16384
echo "== old"; GPIMHIP_OLD_POTRF=1 python tools/potrf_run.py 1280 4224 8192 16384
echo "== new, FILL_CAP=256"; GPIMHIP_FILL_CAP=256 python tools/potrf_run.py 8192 16384
echo "== new, FILL_CAP=64"; GPIMHIP_FILL_CAP=64 python tools/potrf_run.py 8192 16384
for n in 1207 4206; do PROF_STAGES=1 python tests/tools/prof_fit.py $n 40 0 RBF; done
PROF_STAGES=1 python tests/tools/prof_fit.py 16384 6 0 RBF
} > $O/log.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
grep -v amdgpu.ids $O/log.txt
