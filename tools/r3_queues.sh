#!/bin/bash
# does GPU_MAX_HW_QUEUES set by `import gpim_amd` (after `import torch`, before the first GPU call) take effect?
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/queues; rm -rf $O; mkdir -p $O
go() { echo "=== env: $1 | pre: $2" >> $O/log.txt; env $1 SHOW_QUEUES=1 STAGES=1 ITERS=20 timeout 300 python $R/tools/r3_single_ctx.py $2 >> $O/log.txt 2>> $O/err.txt; echo "rc=$?" >> $O/log.txt; }
go "X=0" "c1 streams4"                       # package default (8 queues asked for at import)
go "GPU_MAX_HW_QUEUES=4" "c1 streams4"       # the runtime's own default, explicitly: the slow case must come back
go "X=0" "none"
go "X=0" "c1 c3conc c4 c5conc kron gc"       # the order of bench.py's extras
echo "=== bench.py --workload c3 (timed step: 4 concurrent batches)" >> $O/log.txt
timeout 600 python $R/bench.py --workload c3 --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],1))" >> $O/log.txt
GPU_MAX_HW_QUEUES=4 timeout 600 python $R/bench.py --workload c3 --no-cpu-baseline 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 (4 queues) value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],1))" >> $O/log.txt
cat $O/log.txt; grep -v amdgpu.ids $O/err.txt | tail -8
