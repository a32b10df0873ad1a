#!/bin/bash
# kernel trace of the distributed training iteration on one GPU (P = 1), N = $1
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_dist_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tools/r4_dist_phases.py ${1:-16384} > $O/log.txt 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/summary.txt <<'PY'
import pandas as pd, re, sys
t = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
def short(n):
    m = re.search(r'(\w+)<([^>]*)>\(', n)
    if m: return m.group(1).replace('gemm_tiles_kernel', 'gemm') + '<' + m.group(2).replace(' ', '')[:24] + '>'
    return n.split('(')[0].replace('void ', '')[:48]
t['k'] = t.Kernel_Name.map(short)
t['dur'] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t['gap'] = (t.Start_Timestamp - t.End_Timestamp.shift(1)) / 1e3
km = t.index[t.k.str.startswith('kmat')].tolist()
# the second iteration: from its first kmat launch to the end
starts = [i for n, i in enumerate(km) if n == 0 or i - km[n - 1] > 50]
it = t.loc[starts[-1]:]
print("span ms", (it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e6, len(it), "busy ms", it.dur.sum() / 1e3)
print(it.groupby('k').agg(n=('dur', 'size'), dur_ms=('dur', lambda x: x.sum() / 1e3), avg_us=('dur', 'mean')).sort_values('dur_ms', ascending=False).head(25).to_string())
PY
rm -rf $O/kt
grep -v amdgpu $O/log.txt | tail -9; cat $O/summary.txt
