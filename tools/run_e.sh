cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in base ld0 ld0b0; do
  unset GPIMHIP_PROBE_LD0 GPIMHIP_PROBE_BETA0
  if [ $mode = ld0 ]; then export GPIMHIP_PROBE_LD0=1; fi
  if [ $mode = ld0b0 ]; then export GPIMHIP_PROBE_LD0=1 GPIMHIP_PROBE_BETA0=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$mode -- python $R/tests/tools/prof_fit.py 16384 1 0 Matern52 > /dev/null 2>&1
  f=$(ls -t $R/gpurun_out/kt_$mode/*/*kernel_stats.csv | head -1)
  echo "== $mode"; grep "false, false, 0, 8, 128, 128" $f | cut -c1-110
done
