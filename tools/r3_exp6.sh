#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r3_exp6; mkdir -p $O
{
echo "== new"; python tests/tools/bench_c3.py
echo "== old potrf"; GPIMHIP_OLD_POTRF=1 python tests/tools/bench_c3.py
echo "== new, batch 32"; sed 's/batch=64/batch=32/' tests/tools/bench_c3.py > /tmp/c3b.py; python /tmp/c3b.py
} 2>&1 | grep -v amdgpu.ids > $O/log.txt
cd /tmp; export TMPDIR=/tmp
sed 's/iterations=250/iterations=12/; s/range(2)/range(1)/' $GRAFT_REPO_ROOT/tests/tools/bench_c3.py > /tmp/c3s.py
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python /tmp/c3s.py > /dev/null 2>&1
cat $O/log.txt
