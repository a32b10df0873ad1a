# scratch driver of the last A/B experiment run through gpurun (edit freely; not part of the product)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
PROF_PRECISION=single PROF_STAGES=1 python $R/tests/tools/prof_fit.py 16384 4 65536 Matern52
