#!/bin/bash
o=gpurun_out/r4_exp3; mkdir -p $o
python tools/potrf_run.py 8192 10240 12288 16384 20480 > $o/potrf_new.txt 2>/dev/null
GPIMHIP_OLD_PLAN=1 python tools/potrf_run.py 8192 10240 12288 16384 20480 > $o/potrf_oldplan.txt 2>/dev/null
tail -n 12 $o/*.txt
TAG=newplan bash tools/r4_kt.sh 16384
