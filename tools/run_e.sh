cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_c3 -- python $R/tests/tools/bench_c3.py > /tmp/c3.log 2>&1
tail -2 /tmp/c3.log
f=$(ls -t $R/gpurun_out/kt_c3/*/*kernel_stats.csv | head -1)
head -16 $f | cut -c1-150
