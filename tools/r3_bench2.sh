#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_bench2; mkdir -p $O
python bench.py > $O/c2.json 2> $O/c2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench2/c2.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
for s in d['roofline']['stages']: print(s['stage'], round(s['ms_per_call'],3), round(s['frac'],3), round(s['share_of_step'],3))
print('frac', d['roofline']['frac'], 'chain', d['roofline'].get('chain_us_per_128'))
print(d['cpu_baseline']['sample'])
print({k:(v.get('seconds'), v.get('grid_points_per_s')) for k,v in d.get('extra',{}).items() if isinstance(v,dict)})
print(d.get('rmse_vs_oracle'))
PY
tail -3 $O/c2.err
