#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_dist4
timeout 1500 python -m pytest tests/test_gpu_dist.py "tests/test_gpu_dist2.py::test_two_ranks_one_gpu_product_paths" tests/test_cabi_exports.py -q -x > gpurun_out/r3_dist4/pytest.txt 2>&1
tail -40 gpurun_out/r3_dist4/pytest.txt
