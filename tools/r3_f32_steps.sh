#!/bin/bash
# float step schedule (cholstep32.hip) against the float look-ahead schedule: tests that touch single precision, then times
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/f32steps; rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "single or precision or fp32 or chain_stream" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/log.txt; tail -5 $O/tests.log >> $O/log.txt
go() { echo "=== env: $1 | pre: $2" >> $O/log.txt; env $1 PRECS=single STAGES=1 ITERS=20 MIDN32=1 timeout 300 python $R/tools/r3_single_ctx.py $2 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt; }
go "X=0" "none"
go "GPIMHIP_F32_LOOKAHEAD=1" "none"
go "X=0" "c1 c3conc c4 c5conc kron gc"
go "GPIMHIP_F32_LOOKAHEAD=1" "c1 c3conc c4 c5conc kron gc"
for n in 1280 4224 8192 16384; do
  for e in X=0 GPIMHIP_F32_LOOKAHEAD=1; do
    echo "--- N=$n $e" >> $O/log.txt
    env $e PROF_PRECISION=single PROF_STAGES=1 timeout 300 python $R/tests/tools/prof_fit.py $n 12 0 Matern52 2>> $O/err.txt | grep -v workspace >> $O/log.txt
  done
done
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -8
