"""Launch list of the LAST training iteration in a rocprofv3 --kernel-trace CSV of tests/tools/prof_fit.py (iterations are
delimited by kmat_kernel launches): per kernel name sums, then every launch with workgroups, duration and gap.
usage: r4_kt_iter.py csv [rows] [back]"""
import re, sys
import pandas as pd
t = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 400
def short(n):
    m = re.search(r'(\w+)<([^>]*)>\(', n)
    if m: return m.group(1).replace('gemm_tiles_kernel', 'gemm') + '<' + m.group(2).replace(' ', '')[:24] + '>'
    return n.split('(')[0].replace('void ', '')[:40]
t['k'] = t.Kernel_Name.map(short)
t['dur'] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t['gap'] = (t.Start_Timestamp - t.End_Timestamp.shift(1)) / 1e3
t['wgs'] = t.Grid_Size_X // t.Workgroup_Size_X
km = t.index[t.k.str.startswith('kmat')].tolist()
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # 1: the iteration before the last pair of kmat launches (a prediction follows the fit)
a, b = km[-2 - back], km[-1 - back]
it = t.loc[a:b - 1]
print("iteration span us %.1f  launches %d" % ((it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e3, len(it)))
print(it.groupby('k').agg(n=('dur', 'size'), dur=('dur', 'sum'), avg=('dur', 'mean'), gap=('gap', 'sum')).sort_values('dur', ascending=False).to_string())
pd.set_option('display.width', 200); pd.set_option('display.max_rows', 1000)
print(it[['k', 'wgs', 'dur', 'gap']].head(rows).to_string())
