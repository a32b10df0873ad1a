#!/bin/bash
# round 6, session 5: the look-ahead schedule of the diagonal-block role (one barrier per 16-column step)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s7; rm -rf $O; mkdir -p $O
for v in old2 pipe2 la2; do
  timeout 60 ./tools/potf2_prof_$v > $O/prof_$v.txt 2>&1; echo "== $v rc=$?"; grep "rep 2\|max |\|hash" $O/prof_$v.txt | cut -c1-220
done
(time timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_single.py -x -q) > $O/tests_a.log 2>&1
tail -3 $O/tests_a.log
AB_T=40 timeout 900 bash tools/ab/run5.sh > $O/ab.log 2>&1; grep -v amdgpu $O/ab.log
cd /tmp; export TMPDIR=/tmp
for spec in "4212 30 RBF" "16384 3 Matern52"; do
  set -- $spec
  rm -rf $O/kt_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1 -- python $GRAFT_REPO_ROOT/tests/tools/prof_fit.py $1 $2 0 $3 > $O/kt_$1.log 2>&1
  f=$(find $O/kt_$1 -name '*kernel_stats.csv' | head -1)
  cp $f $O/kstats_$1.csv
  rm -rf $O/kt_$1
done
