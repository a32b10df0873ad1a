"""Queue ids and inter-launch gaps of the factorisation chain (chol_step / panel_solve / diag_update launches) of the
LAST N = 16384 factorisation in a rocprofv3 --kernel-trace CSV.   usage: r4_queue_gaps.py csv"""
import sys
import numpy as np, pandas as pd
t = pd.read_csv(sys.argv[1]).sort_values('Start_Timestamp').reset_index(drop=True)
print("queues used by the process: ", dict(t.groupby('Queue_Id').size()))
if 'Stream_Id' in t.columns: print("streams: ", dict(t.groupby('Stream_Id').size()))
chain = t[t.Kernel_Name.str.contains('chol_step_kernel|panel_solve_kernel|diag_update_kernel')]
big = chain[chain.Kernel_Name.str.contains('chol_step')]
# the last factorisation: the last 128 step launches (plus the few launches after the last step)
idx = big.index[-135:]
c = chain.loc[idx[0]:]
print("chain launches analysed: %d on queue(s) %s%s" % (len(c), sorted(c.Queue_Id.unique()),
      (" stream(s) %s" % sorted(c.Stream_Id.unique())) if 'Stream_Id' in c.columns else ""))
gap = (c.Start_Timestamp.values[1:] - c.End_Timestamp.values[:-1]) / 1e3
dur = (c.End_Timestamp.values - c.Start_Timestamp.values) / 1e3
print("span %.2f ms, sum of kernel durations %.2f ms, sum of gaps %.2f ms" % ((c.End_Timestamp.max() - c.Start_Timestamp.min()) / 1e6, dur.sum() / 1e3, gap.sum() / 1e3))
qs = np.percentile(gap, [5, 25, 50, 75, 95, 99])
print("gap us: p5 %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f p99 %.1f max %.1f" % (*qs, gap.max()))
h, e = np.histogram(gap, bins=[-1e9, 0.5, 2, 5, 10, 20, 40, 80, 1e9])
print("gap histogram (us): " + ", ".join("%s: %d" % (("<=%.1f" % e[i + 1]) if i < len(h) - 1 else ">80", h[i]) for i in range(len(h))))
other = t[(t.Start_Timestamp >= c.Start_Timestamp.min()) & (t.End_Timestamp <= c.End_Timestamp.max()) & ~t.index.isin(c.index)]
print("other launches inside the window: %d on queues %s" % (len(other), dict(other.groupby('Queue_Id').size())))
# ---- the last complete TRAINING iteration (between two kmat launches with a K^-1 product in between): where does the time go?
km = t.index[t.Kernel_Name.str.contains('kmat_kernel')].tolist()
lau = t.index[t.Kernel_Name.str.contains('gemm_tiles_kernel<true, true')].tolist()
pairs = [(a, b) for a, b in zip(km[:-1], km[1:]) if any(a < l < b for l in lau)]
a, b = pairs[-1]
it = t.loc[a:b - 1].copy()
t0 = it.Start_Timestamp.min()
it['dur'] = (it.End_Timestamp - it.Start_Timestamp) / 1e3
it['k'] = it.Kernel_Name.str.replace(r'\(.*', '', regex=True).str.replace('void ', '').str.slice(0, 44)
print("training iteration: span %.2f ms (kmat to next kmat %.2f ms), %d launches, queues %s" % (
    (it.End_Timestamp.max() - t0) / 1e6, (t.Start_Timestamp[b] - t0) / 1e6, len(it), dict(it.groupby('Queue_Id').size())))
g = it.groupby(['k', 'Queue_Id']).agg(n=('dur', 'size'), dur_ms=('dur', lambda x: x.sum() / 1e3))
print(g.sort_values('dur_ms', ascending=False).head(10).round(3).to_string())
# per queue: busy time and idle gaps inside the iteration
for q, d in it.groupby('Queue_Id'):
    d = d.sort_values('Start_Timestamp')
    gaps = (d.Start_Timestamp.values[1:] - d.End_Timestamp.values[:-1]) / 1e3
    print("queue %s: %d launches, busy %.2f ms, gaps > 1 us: %d totalling %.2f ms (largest %.1f us)" % (
        q, len(d), d.dur.sum() / 1e3, int((gaps > 1).sum()), gaps[gaps > 1].sum() / 1e3, gaps.max() if len(gaps) else 0))
tail = it[~it.Kernel_Name.str.contains('chol_step_kernel|panel_solve_kernel|diag_update_kernel')]
print("launches outside the chain (start offset ms, duration ms, queue):")
for _, r in tail.iterrows():
    print("   %8.3f  %8.3f  q%s  %s" % ((r.Start_Timestamp - t0) / 1e6, r.dur / 1e3, r.Queue_Id, r.k))
