#!/bin/bash
# round 6, session 11: the K^-1 launch at C1 size as 128x64 halves (8 waves) instead of 64x64 quadrants (A/B knob)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/s11; rm -rf $O; mkdir -p $O
for rep in 1 2 3; do
  for N in 2560 4212 6000; do
    echo -n "quadrants N=$N: " >> $O/ab.log; python tests/tools/prof_fit.py $N 40 2>&1 | grep "ms/iter" | tail -1 >> $O/ab.log
    echo -n "halves    N=$N: " >> $O/ab.log; GPIMHIP_AB_HALF=1 python tests/tools/prof_fit.py $N 40 2>&1 | grep "ms/iter" | tail -1 >> $O/ab.log
  done
done
cat $O/ab.log
python tools/r5_c3.py 2>&1 | grep -v amdgpu
GPIMHIP_AB_HALF=1 python tools/r5_c3.py 2>&1 | grep -v amdgpu
