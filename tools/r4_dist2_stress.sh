#!/bin/bash
# two ranks on one GPU, 200 repetitions each, with and without torch's caching allocator
o=gpurun_out/r4_dist2_stress; mkdir -p $o
export GPIM_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/r4_dist2_stress.py 200 2>&1 | grep -v amdgpu.ids | grep "rank" > $o/caching.txt
PYTORCH_NO_CUDA_MEMORY_CACHING=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/r4_dist2_stress.py 200 2>&1 | grep -v amdgpu.ids | grep "rank" > $o/no_caching.txt
tail -n 6 $o/*.txt
