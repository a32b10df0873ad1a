#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_test
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r3_test/pytest.txt 2>&1
tail -15 gpurun_out/r3_test/pytest.txt
