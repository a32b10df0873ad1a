#!/bin/bash
# kernel trace of ONE sparse (VFE) slice of config C5: N = 6400, 534 inducing inputs, a few iterations enqueued directly
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_c5_trace; rm -rf $O; mkdir -p $O
cat > /tmp/c5t.py <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT"); sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
from gpim_amd import dist as gd
from problems import ckpfm_cube
cube4 = ckpfm_cube()
gd.reconstruct_slices(cube4[..., :1], axis=-1, sparse=True, indpoints=512, kernel="RBF", learning_rate=0.05, iterations=4)
PY
GPIMHIP_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python /tmp/c5t.py > $O/log.txt 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
cp $f $O/kt.csv; rm -rf $O/kt
tail -3 $O/log.txt
