#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp4; mkdir -p $O
{
echo "== default (quad<=512 tiles, one-shot diag)"; python tools/potrf_run.py 384 1280 4224 6144 8192
echo "== old diag"; GPIMHIP_OLD_DIAG=1 python tools/potrf_run.py 1280 4224
echo "== quad_max 128, half_max 512"; GPIMHIP_FILL_QUAD_MAX=128 GPIMHIP_FILL_HALF_MAX=512 python tools/potrf_run.py 1280 4224 8192
echo "== quad_max 64, half_max 256"; GPIMHIP_FILL_QUAD_MAX=64 GPIMHIP_FILL_HALF_MAX=256 python tools/potrf_run.py 1280 4224 8192
echo "== quad_max 0, half_max 512"; GPIMHIP_FILL_QUAD_MAX=0 GPIMHIP_FILL_HALF_MAX=512 python tools/potrf_run.py 1280 4224 8192
echo "== quad_max 2048"; GPIMHIP_FILL_QUAD_MAX=2048 python tools/potrf_run.py 4224 8192
for n in 1207 4206; do PROF_STAGES=1 python tests/tools/prof_fit.py $n 40 0 RBF; done
} 2>&1 | grep -v amdgpu.ids > $O/log.txt
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
cat $O/log.txt
