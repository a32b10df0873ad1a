#!/bin/bash
# A/B of step-plan constants (tools/ab/lib_*.so built by tools/ab/build1.sh <name> cholstep -D...) on one box, two rounds
cd "$GRAFT_REPO_ROOT" || exit 1
keep=/tmp/lib_keep.so; cp gpim_amd/libgpimhip.so $keep
for rep in 1 2; do
  for f in tools/ab/lib_*.so; do
    v=$(basename $f .so); v=${v#lib_}
    cp $f gpim_amd/libgpimhip.so
    line="$v rep$rep:"
    for spec in "3300 40" "4212 40" "5200 30" "6000 20" "8192 20" "16384 6"; do
      set -- $spec
      ms=$(python tests/tools/prof_fit.py $1 $2 0 RBF 2>&1 | grep "ms/iter" | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/')
      line="$line N=$1 $ms |"
    done
    echo "$line"
  done
done
cp $keep gpim_amd/libgpimhip.so
