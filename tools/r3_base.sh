#!/bin/bash
# round-3 baseline: stage timings + kernel trace at C1 / C3-slice / C2 sizes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_base
mkdir -p $O
cd $R
for n in 1207 4206 8192 16384; do
  T=40; [ $n -ge 8192 ] && T=6
  PROF_STAGES=1 python tests/tools/prof_fit.py $n $T 0 RBF > $O/fit_$n.log 2>&1
done
python tools/potrf_run.py 1280 4224 8192 16384 > $O/potrf.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_c1 -- python $R/tests/tools/prof_fit.py 4206 20 0 RBF > $O/kt_c1.log 2>&1
cd $R
tail -n 8 $O/fit_*.log $O/potrf.log
