#!/bin/bash
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5_dist_inv_trace; rm -rf $O; mkdir -p $O
timeout 280 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tools/r5_dist_inv_trace.py ${1:-65536} > $O/log.txt 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python - "$f" > $O/summary.txt <<'PY'
import pandas as pd, re, sys
t = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
def short(n):
    m = re.search(r'(\w+)<([^>]*)>\(', n)
    if m: return m.group(1).replace('gemm_tiles_kernel', 'gemm') + '<' + m.group(2).replace(' ', '')[:28] + '>'
    return n.split('(')[0].replace('void ', '')[:48]
t['k'] = t.Kernel_Name.map(short)
t['dur'] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t['gap'] = (t.Start_Timestamp - t.End_Timestamp.shift(1)) / 1e3
t['wgs'] = (t.Grid_Size_X // t.Workgroup_Size_X) * (t.Grid_Size_Y // t.Workgroup_Size_Y)
mk = t.index[t.Kernel_Name.str.contains('cumsum|scan', case=False)].tolist()
it = t.loc[mk[-2] + 1: mk[-1] - 1] if len(mk) >= 2 else t
print("span ms", (it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e6, len(it), "busy ms", it.dur.sum() / 1e3, "gaps ms", it.gap.clip(lower=0).sum() / 1e3)
print(it.groupby('k').agg(n=('dur', 'size'), dur_ms=('dur', lambda x: x.sum() / 1e3), avg_us=('dur', 'mean'), gap_ms=('gap', lambda x: x.clip(lower=0).sum() / 1e3)).sort_values('dur_ms', ascending=False).head(25).to_string())
pd.set_option('display.width', 200); pd.set_option('display.max_rows', 500)
n = len(it)
print(it[['k', 'wgs', 'dur', 'gap']].iloc[n // 2: n // 2 + 40].to_string())
PY
rm -rf $O/kt
tail -2 $O/log.txt; cat $O/summary.txt
