#!/bin/bash
# packed-triangle potf2 role (79 KB, two workgroups per CU): same bits? timings with different hosting policies
mkdir -p gpurun_out/r4_exp1
o=gpurun_out/r4_exp1
for n in 1207 4212 8192 16384; do python tools/r3_potrf_hash.py $n 2>/dev/null; done > $o/hash.txt
python tools/potrf_run.py 1280 4224 8192 12288 16384 > $o/potrf_default.txt 2>/dev/null
GPIMHIP_FILL_CAP=100000 GPIMHIP_PAIR=0 python tools/potrf_run.py 8192 12288 16384 > $o/potrf_all_nopair.txt 2>/dev/null
GPIMHIP_FILL_CAP=100000 GPIMHIP_PAIR=1 python tools/potrf_run.py 8192 12288 16384 > $o/potrf_all_pair.txt 2>/dev/null
GPIMHIP_FILL_CAP=256 GPIMHIP_PAIR=0 python tools/potrf_run.py 8192 12288 16384 > $o/potrf_256_nopair.txt 2>/dev/null
GPIMHIP_FILL_CAP=504 GPIMHIP_PAIR=0 python tools/potrf_run.py 8192 12288 16384 > $o/potrf_504_nopair.txt 2>/dev/null
tail -n 20 $o/*.txt
