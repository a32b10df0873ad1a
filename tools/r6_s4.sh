#!/bin/bash
# round 6, session 4: fused finalize with device-scope stores instead of a release fence
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s4; rm -rf $O; mkdir -p $O
(time timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_regimes.py -x -q) > $O/tests_a.log 2>&1
tail -3 $O/tests_a.log
for rep in 1 2; do
  for N in 1207 4212 8192 16384; do
    T=60; [ $N -ge 8192 ] && T=12; [ $N -ge 16384 ] && T=6
    for mode in fused two; do
      echo -n "$mode N=$N rep=$rep: " >> $O/ab.log
      case $mode in
        fused) python tests/tools/prof_fit.py $N $T 2>&1 | grep "ms/iter" | tail -1 >> $O/ab.log;;
        two) GPIMHIP_NO_FUSED_FINALIZE=1 python tests/tools/prof_fit.py $N $T 2>&1 | grep "ms/iter" | tail -1 >> $O/ab.log;;
      esac
    done
  done
done
cat $O/ab.log
python tools/r5_c3.py > $O/c3.log 2>&1; tail -3 $O/c3.log
GPIMHIP_NO_FUSED_FINALIZE=1 python tools/r5_c3.py > $O/c3_two.log 2>&1; tail -3 $O/c3_two.log
./tools/potf2_prof > $O/potf2_prof.txt 2>&1; head -3 $O/potf2_prof.txt
python tools/r6_fused_bits.py > $O/bits.txt 2>&1; tail -8 $O/bits.txt
cd /tmp; export TMPDIR=/tmp
for spec in "4212 30 RBF" "16384 3 Matern52" "1207 30 RBF"; do
  set -- $spec
  rm -rf $O/kt_$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1 -- python $GRAFT_REPO_ROOT/tests/tools/prof_fit.py $1 $2 0 $3 > $O/kt_$1.log 2>&1
  f=$(find $O/kt_$1 -name '*kernel_stats.csv' | head -1)
  cp $f $O/kstats_$1.csv
  rm -rf $O/kt_$1
done
