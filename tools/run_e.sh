R=$GRAFT_REPO_ROOT
for v in 0 700 1500 3000; do export GPIMHIP_MID_TILES=$v; echo "== mid $v"
for n in 2048 4206 6000; do PROF_STAGES=1 python $R/tests/tools/prof_fit.py $n 20 0 RBF 2>&1 | grep -E "stage" | tr '\n' ' '; echo; done; done
