#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_exp5; mkdir -p $O
{
for tm in 256 600 1200 2400; do
echo "== TILE64_MAX=$tm"
for n in 1207 2500 4206 6000; do GPIMHIP_TILE64_MAX=$tm PROF_STAGES=1 python tests/tools/prof_fit.py $n 30 0 RBF | grep -v workspace; done
done
} 2>&1 | grep -v amdgpu.ids > $O/log.txt
cat $O/log.txt
