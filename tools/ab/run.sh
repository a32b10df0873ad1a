#!/bin/bash
# A/B of several builds of the library on ONE box: alternating runs of the same fit (tools/ab/lib_<name>.so)
keep=/tmp/lib_keep.so; cp gpim_amd/libgpimhip.so $keep
for rep in 1 2 3; do
  for f in tools/ab/lib_*.so; do
    v=$(basename $f .so); v=${v#lib_}
    cp $f gpim_amd/libgpimhip.so
    echo "== $v $rep"; PROF_STAGES=1 python tests/tools/prof_fit.py ${AB_N:-16384} ${AB_T:-8} ${AB_M:-65536} Matern52 2>&1 | grep -E "ms/iter|stage|predict" | tail -5
  done
done
cp $keep gpim_amd/libgpimhip.so
