#!/bin/bash
# per-launch durations of the step launches of ONE fit iteration at N = 16384 joined with the launch plan's content
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_c2_steps; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tests/tools/prof_fit.py ${1:-16384} 3 0 Matern52 > $O/log.txt 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python - "$f" ${1:-16384} > $O/steps.txt <<'PY'
import sys, re, ctypes
import numpy as np, pandas as pd
sys.path.insert(0, __import__('os').environ['GRAFT_REPO_ROOT'])
from gpim_amd import _lib
lib = _lib.load()
N = int(sys.argv[2]); nb = (N + 127) // 128
n = ctypes.c_int64(); lib.gpimhip_step_plan_host(nb, 1, None, 0, ctypes.byref(n))
buf = np.zeros((n.value, 6), dtype=np.int32)
lib.gpimhip_step_plan_host(nb, 1, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n.value, ctypes.byref(n))
t = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
t['dur'] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t['wgs'] = t.Grid_Size_X // t.Workgroup_Size_X
km = t.index[t.Kernel_Name.str.contains('kmat_kernel')].tolist()
it = t.loc[km[-2]:km[-1] - 1]
st = it[it.Kernel_Name.str.contains('chol_step_kernel')].reset_index(drop=True)
fd = it[it.Kernel_Name.str.contains('panel_solve|diag_update')]
print("step launches %d: %.2f ms; F+D %.2f ms; iteration %.2f ms" % (len(st), st.dur.sum() / 1e3, fd.dur.sum() / 1e3, (it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e6))
for j in range(len(st)):
    r = buf[buf[:, 0] == j]
    d = (r[:, 4] - r[:, 3]) if len(r) else np.zeros(0)
    upd = r[:, 5] == 0 if len(r) else np.zeros(0, bool)
    kb = int(d.sum()); flop = kb * 2 * 128 ** 3
    shape = re.search(r'chol_step_kernel<([^>]*)>', st.Kernel_Name[j]).group(1).replace(' ', '')
    print("%3d %-20s wgs %5d  %8.1f us  tiles %5d (upd %5d inv %5d) kblocks %6d maxd %2d  %.1f TFLOP/s" % (
        j, shape, st.wgs[j], st.dur[j], len(r), int(upd.sum()), int((~upd).sum()), kb, int(d.max()) if len(d) else 0, flop / st.dur[j] / 1e6))
PY
rm -rf $O/kt; head -150 $O/steps.txt
