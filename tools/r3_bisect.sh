#!/bin/bash
# fp32 regression bisect (C2_single_precision 4.54 -> 5.90 s between 2659073 and HEAD): per-stage times of a
# single-precision fit at N = 16384 for the library built at each candidate commit (tools/_bisect/<hash>) and HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/bisect; rm -rf $O; mkdir -p $O
for prec in single double; do
  for t in tools/_bisect/2659073 tools/_bisect/6479819 tools/_bisect/39f4299 tools/_bisect/c1beaa1 .; do
    [ $prec = double ] && [ $t != tools/_bisect/2659073 ] && [ $t != . ] && continue
    name=$(basename $t); [ $t = . ] && name=HEAD
    echo "=== $name $prec" >> $O/bisect.log
    PROF_PRECISION=$prec PROF_STAGES=1 timeout 300 python $R/tools/r3_bisect_run.py $R/$t 16384 4 0 Matern52 >> $O/bisect.log 2> $O/${name}_$prec.err; echo "rc=$?" >> $O/bisect.log
  done
done
cat $O/bisect.log
