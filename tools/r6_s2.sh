#!/bin/bash
# round 6, session 2: the fused gradient contraction + finalize step, the chunked triangular mat-vec: parity tests, A/B
# against the two-launch path (GPIMHIP_NO_FUSED_FINALIZE), per-kernel stats at C1 / C2 / C3 sizes
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s2; rm -rf $O; mkdir -p $O
(time timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py tests/test_gpu_highprec.py tests/test_gpu_single.py -x -q) > $O/tests_a.log 2>&1
tail -4 $O/tests_a.log
(time timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -x -q) > $O/tests_b.log 2>&1
tail -4 $O/tests_b.log
for rep in 1 2; do
  for N in 1207 2560 4212 8192 16384; do
    T=60; [ $N -ge 8192 ] && T=12; [ $N -ge 16384 ] && T=6
    echo "== fused N=$N rep=$rep" >> $O/ab.log
    python tests/tools/prof_fit.py $N $T 2>&1 | grep "ms/iter" | tail -1 >> $O/ab.log
    echo "== two-launch N=$N rep=$rep" >> $O/ab.log
    GPIMHIP_NO_FUSED_FINALIZE=1 python tests/tools/prof_fit.py $N $T 2>&1 | grep "ms/iter" | tail -1 >> $O/ab.log
  done
done
cat $O/ab.log
python tools/r5_c3.py > $O/c3.log 2>&1; tail -5 $O/c3.log
GPIMHIP_NO_FUSED_FINALIZE=1 python tools/r5_c3.py > $O/c3_old.log 2>&1; tail -5 $O/c3_old.log
cd /tmp; export TMPDIR=/tmp
for spec in "4212 30 RBF" "16384 3 Matern52" "1207 30 RBF"; do
  set -- $spec
  rm -rf $O/kt_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1 -- python $GRAFT_REPO_ROOT/tests/tools/prof_fit.py $1 $2 0 $3 > $O/kt_$1.log 2>&1
  f=$(find $O/kt_$1 -name '*kernel_stats.csv' | head -1)
  cp $f $O/kstats_$1.csv
  rm -rf $O/kt_$1
  echo "---- N=$1"; grep -i "gemv\|grad_reduce\|kmat\|trmv\|finalize\|theta" $O/kstats_$1.csv | cut -c1-60,200-400 | sed 's/"[^"]*"//' 
done
