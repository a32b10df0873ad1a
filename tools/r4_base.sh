#!/bin/bash
# baseline hashes + timings of gpimhip_potrf (round 4 start)
mkdir -p gpurun_out/r4_base
for n in 1207 4212 8192 16384; do python tools/r3_potrf_hash.py $n; done > gpurun_out/r4_base/hash.txt 2>&1
python tools/potrf_run.py 1280 4224 8192 12288 16384 > gpurun_out/r4_base/potrf.txt 2>&1
cat gpurun_out/r4_base/hash.txt gpurun_out/r4_base/potrf.txt
