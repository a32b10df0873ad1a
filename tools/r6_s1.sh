#!/bin/bash
# round 6, session 1: GPU suite on the round's first commit + A/B of the pair rule (plan_inverse's own vs round 5's overwrite)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s1; mkdir -p $O
(time python -m pytest tests -m gpu -x -q) > $O/gputests.log 2>&1
for rep in 1 2 3; do
  for N in 4212 2560 8192 16384; do
    T=40; [ $N -ge 8192 ] && T=12; [ $N -ge 16384 ] && T=6
    echo "== new N=$N rep=$rep" >> $O/ab_pair.log
    python tests/tools/prof_fit.py $N $T 2>&1 | grep "ms/iter" | tail -1 >> $O/ab_pair.log
    echo "== old N=$N rep=$rep" >> $O/ab_pair.log
    GPIMHIP_AB_PAIR_OLD=1 python tests/tools/prof_fit.py $N $T 2>&1 | grep "ms/iter" | tail -1 >> $O/ab_pair.log
  done
done
tail -3 $O/gputests.log; cat $O/ab_pair.log
