#!/bin/bash
cd $GRAFT_REPO_ROOT
{
echo "== batch: step schedule, tiles in their own launch"; python tests/tools/bench_c3.py
echo "== batch: hosted (HOST_MAX_BATCH=1000)"; GPIMHIP_HOST_MAX_BATCH=1000 python tests/tools/bench_c3.py
echo "== old"; GPIMHIP_OLD_POTRF=1 python tests/tools/bench_c3.py
} 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -3
