#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -4
python bench.py --workload c3 --no-cpu-baseline 2>&1 | grep -v amdgpu | tail -1 | cut -c1-700
