#!/bin/bash
# planner: all workgroups of a hosted chunk must end with the launch -- timings at the mid sizes
for n in 1207 2500 4212 6000 8192; do T=40; PROF_STAGES=1 python tests/tools/prof_fit.py $n $T 0 RBF 2>&1 | grep -E "ms/iter|stage" | tail -4 | tr '\n' ' '; echo; done
PROF_STAGES=1 python tests/tools/prof_fit.py 16384 6 0 Matern52 2>&1 | grep -E "ms/iter|stage" | tail -4 | tr '\n' ' '; echo
python tools/r4_c3.py 2>&1 | grep -v amdgpu | head -3
