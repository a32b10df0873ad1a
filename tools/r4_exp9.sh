#!/bin/bash
# ragged last block + split batches: parity tests, then C3 timings
o=gpurun_out/r4_exp9; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_highprec.py tests/test_gpu_fused_predict.py -x -q 2>&1 | tail -8 > $o/tests.txt; cat $o/tests.txt
python tools/r4_c3.py check 2>&1 | grep -v amdgpu.ids
python tools/r4_c3.py 2>&1 | grep -v amdgpu.ids | head -3
PROF_STAGES=1 python tests/tools/prof_fit.py 1207 100 4096 RBF 2>&1 | grep -E "ms/iter|stage|predict"
