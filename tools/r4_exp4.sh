#!/bin/bash
# fused factorisation + inverse: correctness (ops / regimes tests) and iteration times at the three regimes
o=gpurun_out/r4_exp4; mkdir -p $o
python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_e2e.py tests/test_gpu_sparse.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 > $o/tests.txt
PROF_STAGES=1 python tests/tools/prof_fit.py 1207 40 0 RBF > $o/fit1207.txt 2>&1
PROF_STAGES=1 python tests/tools/prof_fit.py 4212 20 0 RBF > $o/fit4212.txt 2>&1
tail -n 6 $o/*.txt
