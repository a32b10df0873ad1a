#!/bin/bash
# does the NT bulk update lose speed at one workgroup per CU (the occupancy it would have inside a kernel that
# also hosts the 134 KB potf2 role)?  in-order schedule (no look-ahead) isolates kernel time.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp1; mkdir -p $O
echo "== baseline look-ahead" > $O/log.txt
python tools/potrf_run.py 16384 >> $O/log.txt 2>&1
echo "== in-order, 2 WG/CU" >> $O/log.txt
GPIMHIP_LOOKAHEAD_MIN_PANELS=1000 python tools/potrf_run.py 8192 16384 >> $O/log.txt 2>&1
echo "== in-order, 1 WG/CU (MID_TILES huge)" >> $O/log.txt
GPIMHIP_LOOKAHEAD_MIN_PANELS=1000 GPIMHIP_MID_TILES=10000000 python tools/potrf_run.py 8192 16384 >> $O/log.txt 2>&1
cd /tmp; export TMPDIR=/tmp
GPIMHIP_LOOKAHEAD_MIN_PANELS=1000 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt2 -- python $GRAFT_REPO_ROOT/tools/potrf_run.py 16384 > /dev/null 2>&1
GPIMHIP_LOOKAHEAD_MIN_PANELS=1000 GPIMHIP_MID_TILES=10000000 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt1 -- python $GRAFT_REPO_ROOT/tools/potrf_run.py 16384 > /dev/null 2>&1
cat $GRAFT_REPO_ROOT/$O/log.txt
