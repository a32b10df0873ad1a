"""The symmetry-reduced exact GP (reconstructor(structured=True), Matern52) on a complete side x side image: seconds per
Adam iteration and per prediction on the full grid; at side <= 128 also against the dense HIP path.   usage: r5_symm_time.py side [T]"""
import sys, os, time
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from problems import lattice_image
side = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 5
R, _ = lattice_image(size=side, frac=1.0, seed=1)
X = gpim.utils.get_full_grid(R)
kw = dict(kernel="Matern52", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, verbose=0)
rec = gpim.reconstructor(X, R, X, structured=True, iterations=2, **kw)
rec.train()
torch.cuda.synchronize(); t = time.perf_counter()
rec.train(iterations=T)
torch.cuda.synchronize(); dt = time.perf_counter() - t
torch.cuda.synchronize(); t = time.perf_counter()
mean, sd = rec.predict()
torch.cuda.synchronize(); dp = time.perf_counter() - t
N = side * side
print("symmetry-reduced %dx%d (N = %d, 4 blocks of %d): %.3f s per Adam iteration (dense-equivalent %.1f TFLOP/s over N^3), predict %.3f s; loss %s"
      % (side, side, N, N // 4, dt / T, float(N) ** 3 / (dt / T) / 1e12, dp, np.round(rec.loss_all[-3:], 4)))
print("workspace GiB %.1f" % (rec._handle.lib.gpimhip_workspace_bytes(rec._handle.h) / 2 ** 30))
if side <= 128:
    dn = gpim.reconstructor(X, R, X, iterations=2, **kw)
    dn.train()
    torch.cuda.synchronize(); t = time.perf_counter()
    dn.train(iterations=T)
    torch.cuda.synchronize(); dd = time.perf_counter() - t
    md, sdd = dn.predict()
    print("dense path: %.3f s per Adam iteration; max |mean - dense| %.2e, max |sd - dense| %.2e, loss %s"
          % (dd / T, np.abs(mean - md).max(), np.abs(sd - sdd).max(), np.round(dn.loss_all[-3:], 4)))
