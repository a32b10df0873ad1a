#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_dist2
timeout 2400 python -m pytest tests/test_gpu_dist2.py -q -m gpu -x > gpurun_out/r3_dist2/pytest.txt 2>&1
tail -40 gpurun_out/r3_dist2/pytest.txt
