#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_bench1; mkdir -p $O
python bench.py --workload c1 --steps 2 --warmup 1 > $O/c1.json 2> $O/c1.err; tail -c 3000 $O/c1.json; tail -3 $O/c1.err
python bench.py --workload c3 --steps 1 --warmup 1 > $O/c3.json 2> $O/c3.err; tail -c 2500 $O/c3.json; tail -3 $O/c3.err
