"""Per-round view of one gpimhip_potrf call from a rocprofv3 --kernel-trace CSV (tools/potrf_run.py):
calls are separated by gaps > 1 ms; for the LAST call of each size prints, per 512-column round, the window of
the panel chain (potf2 -> solve -> in-panel update x4) and of the bulk trailing update, launch gaps included.
usage: potrf_timeline.py <kernel_trace.csv> [detail_round]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "potf2" in n or "gemm_tiles" in n:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"),
                   int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))))
ev.sort()
calls, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > 1_000_000:
        calls.append(cur); cur = []
    cur.append(e)
calls.append(cur)
detail = int(sys.argv[2]) if len(sys.argv) > 2 else -1
seen = {}
for c in calls:
    npot = sum(1 for e in c if "potf2" in e[2])
    seen[npot] = c
for npot, c in sorted(seen.items()):
    t0 = c[0][0]
    tot = (max(e[1] for e in c) - t0) / 1e6
    print("== call with %d potf2 launches (N = %d): %.3f ms, %d launches" % (npot, npot * 128, tot, len(c)))
    pot = [e for e in c if "potf2" in e[2]]
    ksum = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, q, w in c:
        key = ("potf2" if "potf2" in n else n[n.index("<"):n.index(">") + 1]) + " q" + q
        ksum[key][0] += 1; ksum[key][1] += (e - s) / 1e6
    for k, v in sorted(ksum.items(), key=lambda kv: -kv[1][1]):
        print("   %-40s n=%5d  sum %.3f ms  avg %.1f us" % (k, v[0], v[1], v[1] / v[0] * 1e3))
    # rounds: 4 potf2 each
    print("   round: chain window [first potf2 start .. last chain kernel end], potf2 avg us, biggest launch in round (wgs, ms)")
    for r in range(0, len(pot), 4):
        grp = pot[r:r + 4]
        w0 = grp[0][0]
        w1 = pot[r + 4][0] if r + 4 < len(pot) else max(e[1] for e in c)
        inwin = [e for e in c if w0 <= e[0] < w1]
        big = max(inwin, key=lambda e: e[4])
        print("   r%-3d t=%8.3f  round %.3f ms  potf2 avg %5.1f us  nlaunch %3d  biggest: wgs=%-6d %.3f ms (start +%.3f)" % (
            r // 4, (w0 - t0) / 1e6, (w1 - w0) / 1e6, sum(e[1] - e[0] for e in grp) / len(grp) / 1e3, len(inwin), big[4],
            (big[1] - big[0]) / 1e6, (big[0] - w0) / 1e6))
        if r // 4 == detail:
            for s, e, n, q, w in inwin:
                print("        %9.3f %9.3f (%6.1f us) q%-2s wgs=%-6d %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, q, w,
                      "potf2" if "potf2" in n else n[n.index("<"):n.index(">") + 1]))
