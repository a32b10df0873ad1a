#!/bin/bash
# D_j updates the next diagonal tile only (older window columns hosted): parity tests + timings at the three regimes
o=gpurun_out/r4_exp12; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_sparse.py -x -q 2>&1 | tail -4
python tools/r4_c3.py 2>&1 | grep -v amdgpu | head -3
PROF_STAGES=1 python tests/tools/prof_fit.py 4212 60 0 RBF 2>&1 | grep -E "ms/iter|stage" | tail -4
PROF_STAGES=1 python tests/tools/prof_fit.py 16384 6 0 Matern52 2>&1 | grep -E "ms/iter|stage" | tail -4
python tools/r3_c5.py 2>&1 | grep "concurrency 4"
