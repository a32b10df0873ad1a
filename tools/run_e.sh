R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export GPIMHIP_LL_WINDOW=0
for mp in 12 4; do echo "== lookahead min panels $mp"; GPIMHIP_LOOKAHEAD_MIN_PANELS=$mp python $R/tools/potrf_run.py 2048 4096 6144 | grep potrf; done
