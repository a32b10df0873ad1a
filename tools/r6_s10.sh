#!/bin/bash
# round 6, session 10: hand-written exp / sqrt of the covariance functions (kfun.hpp), interior fast paths
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s10; rm -rf $O; mkdir -p $O
timeout 120 ./tools/r6_kfun_probe > $O/kfun_probe.txt 2>&1; cat $O/kfun_probe.txt
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/gputests.log 2>&1; tail -5 $O/gputests.log
for N in 1207 4212 16384; do
  T=60; [ $N -ge 16384 ] && T=6
  python tests/tools/prof_fit.py $N $T 0 $([ $N -ge 16384 ] && echo Matern52 || echo RBF) 2>&1 | grep "ms/iter" | tail -1
done
python tools/r5_c3.py 2>&1 | grep -v amdgpu
cd /tmp; export TMPDIR=/tmp
for spec in "4212 30 RBF" "16384 3 Matern52" "1207 30 RBF"; do
  set -- $spec
  rm -rf $O/kt_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1 -- python $GRAFT_REPO_ROOT/tests/tools/prof_fit.py $1 $2 0 $3 > $O/kt_$1.log 2>&1
  f=$(find $O/kt_$1 -name '*kernel_stats.csv' | head -1)
  cp $f $O/kstats_$1.csv
  rm -rf $O/kt_$1
  echo "--- N=$1"; grep "grad_reduce\|kmat_kernel\|gemv\|trmv\|finalize" $O/kstats_$1.csv | awk -F'","' '{printf "%s calls %s avg %.1f us\n", substr($1,2,45), $2, $4/1000}'
done
