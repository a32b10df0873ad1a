#!/bin/bash
# VGPR accumulators in the one-shot chain kernels and the fused predictor (were AGPRs: half-rate fp64 MFMA issue)
o=gpurun_out/r4_exp11; mkdir -p $o
for n in 1207 4212 8192; do python tools/r3_potrf_hash.py $n 2>&1 | grep sha; done
TAG=vgpr bash tools/r4_kt_fit.sh 4212 12 RBF > /dev/null 2>&1; head -14 gpurun_out/r4_ktfit_vgpr/iter.txt
PROF_STAGES=1 python tests/tools/prof_fit.py 4212 60 0 RBF 2>&1 | grep -E "ms/iter|stage"
python tests/tools/bench_bo_large.py 2>&1 | tail -5
