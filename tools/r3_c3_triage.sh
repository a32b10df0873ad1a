#!/bin/bash
# Triage of the rocprofv3 SIGSEGV seen in `bench.py --workload c3` (see tools/r3_c3_triage.py).  Every run is its own
# process under its own timeout; logs, /proc/self/maps and the crash reports land in gpurun_out/c3tri.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c3tri; rm -rf $O; mkdir -p $O
T=$R/tools/r3_c3_triage.py
PROF="rocprofv3 --kernel-trace --stats --output-format csv"
run() {  # run <tag> <command...>
  tag=$1; shift
  timeout 300 "$@" > $O/$tag.log 2> $O/$tag.err; echo "$tag rc=$?" >> $O/summary.txt
}
# A: the failing configuration (4 concurrent lock-step batches of 16, captured iterations), three processes
for k in 1 2 3; do run profA$k $PROF -d $O/profA$k -- python $T profA$k 16 4 2; done
# B: the same without graph capture (plain launches from four host threads)
for k in 1 2; do GPIMHIP_NO_GRAPH=1 run profB$k $PROF -d $O/profB$k -- python $T profB$k 16 4 2; done
# C: one lock-step batch of 64 on one stream / handle (captured iterations, no host threads)
run profC1 $PROF -d $O/profC1 -- python $T profC1 64 1 2
# D: no profiler: the concurrent configuration repeated, every repetition compared bitwise with the first, and the
#    serial result for the cross-check of D against C on the dev box
run plainD1 python $T plainD1 16 4 8
run plainD2 python $T plainD2 16 4 8
run plainS1 python $T plainS1 16 1 2
find $O -name "*_kernel_trace.csv" -delete
cat $O/summary.txt
