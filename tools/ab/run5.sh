#!/bin/bash
# A/B of the builds tools/ab/lib_*.so on ONE box, alternating: C1-size and N = 8192 iterations, config C3 (16 x 4) and its
# per-rank share (8 slices, batch='auto')
keep=/tmp/lib_keep.so; cp gpim_amd/libgpimhip.so $keep
for rep in 1 2; do
  for f in tools/ab/lib_*.so; do
    v=$(basename $f .so); v=${v#lib_}
    cp $f gpim_amd/libgpimhip.so
    echo "== $v $rep"
    for n in ${AB_SIZES:-4212 8192}; do python tests/tools/prof_fit.py $n ${AB_T:-30} 0 RBF 2>&1 | grep "ms/iter" | tail -1; done
    [ -z "$AB_NO_C3" ] && python tools/r5_c3.py
  done
done
cp $keep gpim_amd/libgpimhip.so
