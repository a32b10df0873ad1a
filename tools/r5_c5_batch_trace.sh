#!/bin/bash
# kernel trace of one Adam iteration of config C5's five sparse slices as ONE lock-step batch (gpimhip_fit_vfe_batched)
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_c5_batch_trace; rm -rf $O; mkdir -p $O
cat > /tmp/c5b.py <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT"); sys.path.insert(0, "$GRAFT_REPO_ROOT/tests")
from gpim_amd import dist as gd
from problems import ckpfm_cube
cube4 = ckpfm_cube()
gd.reconstruct_slices(cube4, axis=-1, sparse=True, indpoints=512, kernel="RBF", learning_rate=0.05, iterations=4)
PY
GPIMHIP_NO_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python /tmp/c5b.py > $O/log.txt 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
cp $f $O/kt.csv; rm -rf $O/kt
python - <<PY
import pandas as pd, re
t = pd.read_csv("$O/kt.csv").sort_values("Start_Timestamp").reset_index(drop=True)
t["dur"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t["wgs"] = (t.Grid_Size_X // t.Workgroup_Size_X) * (t.Grid_Size_Y // t.Workgroup_Size_Y)
t["k"] = t.Kernel_Name.map(lambda n: re.sub(r"\(.*", "", n.replace("void ", ""))[:64])
idx = t.index[t.k == "theta_kernel_strided"].tolist()
it = t.loc[idx[-2]:idx[-1] - 1]
print("iteration span us %.1f, kernels %d, sum dur %.1f" % ((it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e3, len(it), it.dur.sum()))
pd.set_option("display.width", 250); pd.set_option("display.max_rows", 500)
print(it.groupby("k").agg(n=("dur", "size"), dur=("dur", "sum"), avg=("dur", "mean")).sort_values("dur", ascending=False).to_string())
print(it[["k", "wgs", "dur"]].to_string())
PY
