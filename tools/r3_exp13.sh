#!/bin/bash
cd $GRAFT_REPO_ROOT
{
for mt in 2048 1100 700; do echo "== MID_TILES=$mt"; GPIMHIP_MID_TILES=$mt PROF_STAGES=1 python tests/tools/prof_fit.py 16384 4 0 Matern52 | grep -v workspace; done
echo "== MID_TILES=1100 at 8192"; GPIMHIP_MID_TILES=1100 PROF_STAGES=1 python tests/tools/prof_fit.py 8192 6 0 Matern52 | grep -v workspace
echo "== default at 8192"; PROF_STAGES=1 python tests/tools/prof_fit.py 8192 6 0 Matern52 | grep -v workspace
} 2>&1 | grep -v "amdgpu"
