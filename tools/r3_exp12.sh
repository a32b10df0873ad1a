#!/bin/bash
cd $GRAFT_REPO_ROOT
{
for pm in 0 1; do echo "== PAIR=$pm"; GPIMHIP_PAIR=$pm python tools/potrf_run.py 4224 8192 12288 16384 20480; done
echo "== PAIR=1 cap 0"; GPIMHIP_PAIR=1 GPIMHIP_FILL_CAP=0 python tools/potrf_run.py 8192 16384
echo "== PAIR=1 cap 128"; GPIMHIP_PAIR=1 GPIMHIP_FILL_CAP=128 python tools/potrf_run.py 8192 16384
} 2>&1 | grep -v "amdgpu"
