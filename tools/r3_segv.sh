#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/segv1 -- python $R/tests/tools/prof_fit.py 16384 2 0 Matern52 > /tmp/o1.txt 2>&1; echo "prof_fit 16384 with close: rc $?"; tail -3 /tmp/o1.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/segv2 -- python $R/tests/tools/prof_fit.py 8192 2 0 Matern52 > /tmp/o2.txt 2>&1; echo "prof_fit 8192: rc $?"
GPIMHIP_LOOKAHEAD_MIN_PANELS=1000 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/segv3 -- python $R/tests/tools/prof_fit.py 16384 2 0 Matern52 > /tmp/o3.txt 2>&1; echo "prof_fit 16384 no side streams: rc $?"
