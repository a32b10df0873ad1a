#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do timeout 600 python -m pytest "tests/test_gpu_dist2.py::test_two_ranks_one_gpu_product_paths" -q -x 2>&1 | tail -2; done
