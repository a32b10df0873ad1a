cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python /root/repo/tests/tools/prof_fit.py 4096 10 0 Matern52 > /dev/null 2>&1
python3 - <<PY
import csv,glob,re
rows=list(csv.DictReader(open(glob.glob('/tmp/kt/*/*kernel_trace.csv')[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
pot=[i for i,r in enumerate(rows) if 'potf2' in r['Kernel_Name']]
# last iteration's potrf = last 32 potf2
i0=pot[-32]; i1=pot[-1]+3
t0=int(rows[i0]['Start_Timestamp'])
def short(n):
    if 'potf2' in n: return 'potf2'
    m=re.search(r'<(.*?)>',n); return (n.split('<')[0][-18:]+'<'+m.group(1)+'>') if m else n[:40]
prev=None
for r in rows[i0:i0+14]+rows[i1-8:i1+1]:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    print("%-44s wgs %5d  start %8.1f dur %6.1f gap %5.1f"%(short(r['Kernel_Name']), int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), s, e-s, s-prev if prev is not None else 0))
    prev=e
print("potrf span %.1f us"%((int(rows[i1]['End_Timestamp'])-t0)/1e3))
PY
