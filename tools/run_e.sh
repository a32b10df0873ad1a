cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in off 96 80 64 48 32; do
  if [ $v = off ]; then unset GPIMHIP_RELAX_ROWS; else export GPIMHIP_RELAX_ROWS=$v; fi
  echo "== relax rows > $v"; PROF_STAGES=1 python $R/tests/tools/prof_fit.py 16384 3 0 Matern52 2>&1 | grep -E "stage potrf"
done
