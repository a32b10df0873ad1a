#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp7; mkdir -p $O
{
echo "== old in-order 8-wave NT"; GPIMHIP_OLD_POTRF=1 GPIMHIP_LOOKAHEAD_MIN_PANELS=1000 python tools/potrf_run.py 16384
echo "== old in-order 4-wave NT"; GPIMHIP_NT_4WAVE=1 GPIMHIP_OLD_POTRF=1 GPIMHIP_LOOKAHEAD_MIN_PANELS=1000 python tools/potrf_run.py 16384
echo "== old lookahead 4-wave NT"; GPIMHIP_NT_4WAVE=1 GPIMHIP_OLD_POTRF=1 python tools/potrf_run.py 16384
echo "== step schedule all N, FILL_CAP 0 (bulk standalone 8-wave)"; GPIMHIP_STEP_MAX_NP=100000 GPIMHIP_FILL_CAP=0 python tools/potrf_run.py 16384
echo "== step schedule all N, FILL_CAP 0, 4-wave NT"; GPIMHIP_NT_4WAVE=1 GPIMHIP_STEP_MAX_NP=100000 GPIMHIP_FILL_CAP=0 python tools/potrf_run.py 16384
} 2>&1 | grep -v amdgpu.ids > $O/log.txt
cat $O/log.txt
