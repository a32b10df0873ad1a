#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp8; mkdir -p $O
{
for cap in 0 32 128 100000; do
echo "== step schedule all N, FILL_CAP $cap"; GPIMHIP_NT_4WAVE=1 GPIMHIP_STEP_MAX_NP=100000 GPIMHIP_FILL_CAP=$cap python tools/potrf_run.py 8192 10240 12288 16384 20480
done
echo "== old lookahead"; GPIMHIP_OLD_POTRF=1 python tools/potrf_run.py 10240 12288 20480
} 2>&1 | grep -v "amdgpu.ids\|residual" > $O/log.txt
cat $O/log.txt
