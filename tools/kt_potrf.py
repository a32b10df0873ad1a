"""Launch list of the LAST potrf call of a given order in a rocprofv3 --kernel-trace CSV of tools/potrf_run.py.
usage: kt_potrf.py <kernel_trace.csv> <nb> [rows]"""
import re, sys
import pandas as pd
t = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
nb = int(sys.argv[2]); rows = int(sys.argv[3]) if len(sys.argv) > 3 else 40
def short(n):
    m = re.search(r'(\w+)<([^>]*)>\(', n)
    if m: return m.group(1).replace('gemm_tiles_kernel', 'gemm') + '<' + m.group(2).replace(' ', '') + '>'
    return n.split('(')[0].replace('void ', '')[:40]
t['k'] = t.Kernel_Name.map(short)
t['dur'] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t['gap'] = (t.Start_Timestamp - t.End_Timestamp.shift(1)) / 1e3
t['wgs'] = t.Grid_Size_X // t.Workgroup_Size_X
isf = t.k.str.startswith('chol_step') | t.k.str.startswith('potf2')
# group factorisations: a run of nb potf2-like launches whose last has the smallest grid
idx = t.index[isf].tolist()
# find last index sequence of length nb ending at a launch with wgs==1 preceded by nb-1 steps
# a factorisation = nb consecutive step launches; its first panel solve has 4 (nb - 1) workgroups
ps = t.index[t.k.str.startswith('panel_solve') & (t.wgs == 4 * (nb - 1))].tolist()
cands = [max(i for i in idx if i < q) for q in ps]
first = [i for i in cands if t.wgs[i] == 1][-1]       # block column 0 hosts no fillers
e0 = idx.index(first)
a, b = idx[e0], idx[e0 + nb - 1]
it = t.loc[a:b]
print("span us %.1f  launches %d" % ((it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e3, len(it)))
print(it.groupby('k').agg(n=('dur', 'size'), dur=('dur', 'sum'), avg=('dur', 'mean'), gap=('gap', 'sum')).sort_values('dur', ascending=False).to_string())
pd.set_option('display.width', 200)
print(it[['k', 'wgs', 'dur', 'gap']].head(rows).to_string())
