#!/bin/bash
# kernel trace of gpimhip_potrf alone at N = $1 with the environment given in $2.. ; output gpurun_out/r4_kt_$TAG
# usage: TAG=name bash tools/r4_kt.sh N [ENV=VAL ...]
cd /tmp; export TMPDIR=/tmp
N=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/r4_kt_${TAG:-x}; rm -rf $O; mkdir -p $O
env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tools/potrf_run.py $N > $O/log.txt 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python $GRAFT_REPO_ROOT/tools/r4_kt_steps.py $f $((($N + 127) / 128)) > $O/steps.txt 2>&1
rm -rf $O/kt
tail -3 $O/log.txt
