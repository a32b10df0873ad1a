"""Per-kernel breakdown and launch list of ONE training iteration (between two theta_kernel launches) from a
rocprofv3 --kernel-trace CSV.   usage: kt_iter.py <kernel_trace.csv> [n_rows_to_list]"""
import re, sys
import pandas as pd
t = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 0
def short(n):
    m = re.search(r'(\w+)<([^>]*)>\(', n)
    if m: return m.group(1).replace('gemm_tiles_kernel', 'gemm') + '<' + m.group(2).replace(' ', '') + '>'
    return n.split('(')[0].replace('void ', '')[:40]
t['k'] = t.Kernel_Name.map(short)
t['dur'] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
idx = t.index[t.k.str.startswith('theta_kernel')].tolist()
a, b = idx[-3], idx[-2]
it = t.loc[a:b - 1]
print("iteration span us %.1f, %d launches" % ((it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e3, len(it)))
print(it.groupby('k').agg(n=('dur', 'size'), dur=('dur', 'sum'), avg=('dur', 'mean')).sort_values('dur', ascending=False).to_string())
if nlist:
    pd.set_option('display.width', 200)
    it = it.assign(wgs=it.Grid_Size_X // it.Workgroup_Size_X)
    print(it[['k', 'wgs', 'Grid_Size_Y', 'Workgroup_Size_X', 'dur']].head(nlist).to_string())
