R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python -m pytest $R/tests/test_gpu_e2e.py $R/tests/test_gpu_select.py $R/tests/test_gpu_fused_predict.py -x -q 2>&1 | tail -4
BO_PROFILE=1 python $R/tests/tools/bench_bo_large.py 2>&1 | tail -26
python $R/tests/tools/bench_bo.py 2>&1 | tail -2
