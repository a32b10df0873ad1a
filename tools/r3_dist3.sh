#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_dist3
timeout 1200 python -m pytest tests/test_gpu_dist.py tests/test_gpu_dist2.py tests/test_cabi_exports.py -q -x > gpurun_out/r3_dist3/pytest.txt 2>&1
tail -30 gpurun_out/r3_dist3/pytest.txt
python bench.py --workload c2full --steps 1 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | grep -v amdgpu | tail -2 | cut -c1-900
