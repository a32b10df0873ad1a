"""Per-iteration time of the distributed training loop (exact_gp_fit) on one GPU.  usage: dist_fit_time.py N [T] [streams4]
streams4: four extra torch streams are created and touched first -- the process state in which round 3 / 4 saw dependent
launches start 30-45 us late when a side stream's pending wait sat in the fifth or a later hardware queue (DESIGN.md
section 6); the distributed driver still has a side stream."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpim_amd.dist_chol import exact_gp_fit
from problems import lattice_image
N = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
side = int(np.sqrt(N))
R, _ = lattice_image(size=side, frac=1.0, seed=1)
ii, jj = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="ij")
X = np.stack([ii.ravel(), jj.ravel()], 1); y = R.ravel()
kw = dict(kernel="Matern52", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1)
if "streams4" in sys.argv[3:]:
    keep = []
    for _ in range(4):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            torch.zeros(1024, device="cuda").add_(1.0)
        st.synchronize(); keep.append(st)
exact_gp_fit(X[:2048], y[:2048], iterations=1, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
hyper, u = exact_gp_fit(X, y, iterations=T, **kw)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("exact_gp_fit N=%d: %.3f s per Adam iteration = %.1f TFLOP/s over N^3; loss %s" % (N, dt / T, float(N) ** 3 / (dt / T) / 1e12, hyper["loss"]))
print("peak memory GiB", torch.cuda.max_memory_allocated() / 2 ** 30)
