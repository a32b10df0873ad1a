#!/bin/bash
# kernel trace of lock-step batches of config C3's slices (N = 1207, nb = 10): $1 slices in one batch, a few iterations
cd /tmp; export TMPDIR=/tmp
B=${1:-4}
O=$GRAFT_REPO_ROOT/gpurun_out/r5_c3_trace_b$B; rm -rf $O; mkdir -p $O
cat > /tmp/c3t.py <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT"); sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=4, verbose=0)
gd.reconstruct_slices(R[..., :$B], axis=-1, batch=$B, batch_concurrency=1, **kw)
PY
GPIMHIP_NO_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python /tmp/c3t.py > $O/log.txt 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/r4_kt_iter.py $f 200 1 > $O/iter.txt 2>&1
rm -rf $O/kt
head -80 $O/iter.txt
