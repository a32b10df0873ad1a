cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/k.py <<'PY'
import sys, numpy as np, torch, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
import gpim_amd as gpim
n = int(sys.argv[2])
shape = (n, n)
idx = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
R = np.prod([np.cos(g / (5.0 + k)) for k, g in enumerate(idx)], axis=0) + 0.05 * np.random.default_rng(0).standard_normal(shape)
Xf = gpim.utils.get_full_grid(R)
rec = gpim.reconstructor(Xf, R, Xf, kernel="RBF", structured=True, learning_rate=0.1, iterations=4, verbose=0, lengthscale=[[1.,1.],[40.,40.]])
rec.train()
PY
for n in 64 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_kron$n -- python /tmp/k.py $R $n > /dev/null 2>&1
f=$(find $R/gpurun_out/kt_kron$n -name "*kernel_stats.csv" | head -1)
echo "== n=$n"; head -8 $f | cut -c1-150
done
