#!/bin/bash
# Collects the rocprofv3 summaries committed under profiles/ (run on the MI355X box through gpurun;
# outputs land in gpurun_out/prof_<round tag> -- delete that directory on the dev box first, gpurun merges into it and
# rocprofv3 names its files by process id -- and are copied into profiles/ by hand, then
# tools/summarize_profiles.py regenerates the derived tables).
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
# 1. kernel stats of the bench command itself (+ the JSON lines with and without the profiler)
# rocprofv3 died twice in 9 runs of the concurrent C3 step (SIGSEGV inside librocprofiler-sdk's queue interceptor, first
# submissions of four host threads; profiles/r03_c3_rocprof_crash.txt): a profiled run that leaves no JSON line is
# repeated (at most 2 attempts, 7 minutes each: round 4 also saw the profiled run HANG for the whole 50-minute limit of a
# gpurun call, no kernel launched) and the failed attempt's log is kept as <name>_profiled.crash<k>.err
profiled() {  # profiled <name> <bench.py arguments...>
  name=$1; shift
  for k in 1 2; do
    rm -rf $O/${name}_stats
    timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_stats -- python $R/bench.py "$@" --no-cpu-baseline > $O/${name}_line_profiled.json 2> $O/${name}_profiled.err
    [ -s $O/${name}_line_profiled.json ] && return 0
    mv $O/${name}_profiled.err $O/${name}_profiled.crash$k.err
  done
  return 1
}
profiled bench
timeout 900 python $R/bench.py > $O/bench_line.json 2> $O/bench.err
# 1b. the configs north_star names as targets: C1 (128x128 spiral scan) and C3 (64 slices of 64x64), kernel stats + lines
for wl in c1 c3; do
  profiled $wl --workload $wl
  timeout 600 python $R/bench.py --workload $wl > $O/${wl}_line.json 2> $O/${wl}.err
done
# 2. PMC passes on two training iterations at the bench size (separate runs, kernel trace only)
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -- python $R/tests/tools/prof_fit.py 16384 2 0 Matern52 > /dev/null 2>&1
done
# kernel traces of the bench runs are tens of MB each (gpurun copies back at most 64 MiB): the stats CSVs are what is kept
find $O/bench_stats $O/c1_stats $O/c3_stats -name "*_kernel_trace.csv" -delete
# the profiled bench also traces its child process (the extras): keep the PARENT's stats (lowest pid)
ls $O/bench_stats/*/*_kernel_stats.csv | sort -t/ -k1 -V | head -3
find $O -name "*.csv" | head -30
