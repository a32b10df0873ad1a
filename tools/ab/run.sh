#!/bin/bash
# A/B of two builds of the library on ONE box: alternating runs of the same fit
for rep in 1 2 3; do
  for v in old new; do
    cp tools/ab/lib_$v.so gpim_amd/libgpimhip.so
    echo "== $v $rep"; PROF_STAGES=1 python tests/tools/prof_fit.py 16384 8 65536 Matern52 2>&1 | grep -E "ms/iter|stage|predict" | tail -6
  done
done
cp tools/ab/lib_new.so gpim_amd/libgpimhip.so
