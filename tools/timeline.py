"""Reads a rocprofv3 --kernel-trace CSV and prints, for the LAST fit iteration in it, the timeline of
launches: per stream (queue) the busy time, and a coarse Gantt of which kernel classes were running.
usage: timeline.py <kernel_trace.csv> [t0_ms t1_ms]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
               int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)))
ev.sort()
def short(n):
    n = n.replace("gemm_tiles_kernel", "gemm")
    return n[:60]
# split into iterations at kmat_kernel launches
starts = [i for i, e in enumerate(ev) if "kmat_kernel" in e[2]]
if len(starts) >= 2:
    a, b = starts[-2], starts[-1]
else:
    a, b = 0, len(ev)
it = ev[a:b]
t0 = it[0][0]
print("iteration: %d launches, %.3f ms" % (len(it), (max(e[1] for e in it) - t0) / 1e6))
byq = collections.defaultdict(float)
for s, e, n, q, g, w in it:
    byq[q] += (e - s) / 1e6
print("busy ms per queue:", dict(byq))
byk = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q, g, w in it:
    k = short(n) + " q" + str(q)
    byk[k][0] += 1; byk[k][1] += (e - s) / 1e6
for k, v in sorted(byk.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-70s n=%4d  %.3f ms" % (k, v[0], v[1]))
# union of busy intervals = time with at least one kernel running
iv = sorted((s, e) for s, e, *_ in it)
cur_s, cur_e, busy = iv[0][0], iv[0][1], 0
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("time with >= 1 kernel running: %.3f ms" % (busy / 1e6))
if len(sys.argv) > 3:
    lo, hi = float(sys.argv[2]), float(sys.argv[3])
    for s, e, n, q, g, w in it:
        if lo <= (s - t0) / 1e6 <= hi:
            print("%9.3f %9.3f  q%-3s wgs=%-6d %s" % ((s - t0) / 1e6, (e - t0) / 1e6, q, g // max(w, 1), short(n)))
