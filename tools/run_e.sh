R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export GPIMHIP_TG_SPIN=2000000
echo "== TG"; timeout 120 python $R/tools/potrf_run.py 2048 4096 6144 8192 16384 2>&1 | grep potrf; echo "rc $?"
