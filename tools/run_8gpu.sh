#!/bin/bash
# The multi-GPU bench lines of one 8-GPU MI355X node, exactly as the driver launches them (one rank per GPU over
# RCCL/xGMI): the weak-scaling headline (c2), the strong-scaling slice workload (c3) and ONE exact GP on the complete
# 256x256 image across the GPUs (c2full), and the whole 64x64x64 cube as one GP with its reflection blocks dealt to the GPUs
# (c3cube).   usage: tools/run_8gpu.sh [N_GPUS=8] [PORT=29555]
# No scaling curve has been measured by the builder: no multi-GPU box is available to gpurun.
N=${1:-8}; PORT=${2:-29555}
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
# ONE exact GP on the complete image across the GPUs: phase times of an Adam iteration per rank (what each rank waited
# for included; the compute-only share of a rank, measured on one GPU: tools/r5_dist_rank_share.py)
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
    tools/r5_dist_phases_mp.py 65536
for w in c2 c3 c2full c3cube; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
      bench.py --gpus $N --workload $w --steps 2 --warmup 1
done
