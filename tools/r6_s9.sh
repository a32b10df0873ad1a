#!/bin/bash
# round 6, session 9: full GPU suite + the default bench command (as the driver runs it)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s9; rm -rf $O; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/gputests.log 2>&1; tail -5 $O/gputests.log
(time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_line.json 2> $O/bench.err; tail -c 1500 $O/bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s9/bench_line.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
for s in d["roofline"]["stages"]: print("  ", s["stage"][:40], round(s["ms_per_call"],3), round(s["frac"],4), {k:round(v,3) for k,v in s.items() if k.startswith("ms_")})
print("chain", d["roofline"]["chain_us_per_128"], d["roofline"]["stages_from"])
e=d.get("extra",{})
for k in ("C1","C3","C3_per_rank_8","C4","C5"):
    v=e.get(k,{}); print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("seconds","ms_per_adam_iteration","mfma_frac","indices_equal_oracle","hyper_max_rel","us_per_adam_iteration")})
    if "roofline" in v:
        for s in v["roofline"]["stages"]: print("     ", s["stage"][:30], round(s["ms_per_call"],3), round(s["frac"],4))
        print("      chain_us_per_128", v["roofline"]["chain_us_per_128"])
print("C4 detail", {k:v for k,v in e.get("C4",{}).items() if k.startswith("hyper_")})
print("rmse headline", d.get("rmse_vs_oracle_headline",{}).get("rmse_mean"), "cpu", d["cpu_baseline"]["value"])
P
