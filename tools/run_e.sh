R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/fp32_probe.py 8192 16384 2>&1 | tail -8
