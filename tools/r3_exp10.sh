#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp10; mkdir -p $O
{
echo "== ring, default caps"; python tools/potrf_run.py 1280 4224 8192 12288 16384
for cap in 0 128 512 100000; do echo "== ring, FILL_CAP $cap"; GPIMHIP_FILL_CAP=$cap python tools/potrf_run.py 8192 16384; done
echo "== ring, quad_max 0 half_max 100000 (halves only), cap 128"; GPIMHIP_FILL_QUAD_MAX=0 GPIMHIP_FILL_HALF_MAX=100000 GPIMHIP_FILL_CAP=128 python tools/potrf_run.py 4224 8192 16384
echo "== ring, quad_max 100000 (quadrants only), cap 128"; GPIMHIP_FILL_QUAD_MAX=100000 GPIMHIP_FILL_CAP=128 python tools/potrf_run.py 4224 8192 16384
echo "== ring, full tiles only, cap 100000"; GPIMHIP_FILL_QUAD_MAX=0 GPIMHIP_FILL_HALF_MAX=0 GPIMHIP_FILL_CAP=100000 python tools/potrf_run.py 8192 16384
} 2>&1 | grep -v "amdgpu.ids\|residual" > $O/log.txt
cat $O/log.txt
