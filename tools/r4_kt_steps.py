"""Per-launch list of the LAST factorisation in a rocprofv3 --kernel-trace CSV of tools/potrf_run.py: kind, workgroups,
duration, gap to the previous launch; per outer panel (4 block columns) the sums.   usage: r4_kt_steps.py csv nb"""
import re, sys
import pandas as pd
t = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
nb = int(sys.argv[2])
def short(n):
    m = re.search(r'(\w+)<([^>]*)>\(', n)
    if m: return m.group(1).replace('gemm_tiles_kernel', 'gemm') + '<' + m.group(2).replace(' ', '') + '>'
    return n.split('(')[0].replace('void ', '')[:40]
t['k'] = t.Kernel_Name.map(short)
t['dur'] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t['gap'] = (t.Start_Timestamp - t.End_Timestamp.shift(1)) / 1e3
t['wgs'] = t.Grid_Size_X // t.Workgroup_Size_X
steps = t.index[t.k.str.startswith('chol_step')].tolist()
last = steps[-nb:]
a, b = last[0], last[-1]
it = t.loc[a:b].copy()
print("span us %.1f  launches %d" % ((it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e3, len(it)))
print(it.groupby('k').agg(n=('dur', 'size'), dur=('dur', 'sum'), avg=('dur', 'mean'), gap=('gap', 'sum')).sort_values('dur', ascending=False).to_string())
it['step'] = it.k.str.startswith('chol_step').cumsum() - 1
# a bulk launch in front of a step belongs to that step's panel
it.loc[it.k.str.startswith('gemm'), 'step'] += 1
it['panel'] = it.step // 4
g = it.groupby('panel')
out = pd.DataFrame({'t0_ms': (g.Start_Timestamp.min() - it.Start_Timestamp.min()) / 1e6,
                    'span_us': (g.End_Timestamp.max() - g.Start_Timestamp.min()) / 1e3,
                    'H_us': it[it.k.str.startswith('chol_step')].groupby('panel').dur.sum(),
                    'H_wgs': it[it.k.str.startswith('chol_step')].groupby('panel').wgs.sum(),
                    'F_us': it[it.k.str.startswith('panel_solve')].groupby('panel').dur.sum(),
                    'D_us': it[it.k.str.startswith('diag_update')].groupby('panel').dur.sum(),
                    'bulk_us': it[it.k.str.startswith('gemm')].groupby('panel').dur.sum(),
                    'gaps_us': g.gap.sum()})
pd.set_option('display.width', 250); pd.set_option('display.max_rows', 500)
print(out.round(1).to_string())
print(it[['k', 'wgs', 'dur', 'gap']].head(40).to_string())
