#!/bin/bash
o=gpurun_out/r4_exp2; mkdir -p $o
./tools/potf2_prof > $o/potf2_prof.txt 2>&1
for n in 1207 4212 8192 16384; do python tools/r3_potrf_hash.py $n 2>/dev/null; done > $o/hash.txt
python tools/potrf_run.py 1280 4224 8192 16384 > $o/potrf_default.txt 2>/dev/null
cat $o/potf2_prof.txt $o/hash.txt $o/potrf_default.txt
