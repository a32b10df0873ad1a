import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
for nsl in (8, 16, 32):
    Rs = R[..., :nsl]
    for batch, conc in ((nsl, 1), (nsl // 2, 2), (nsl // 4, 4), (max(1, nsl // 8), 8)):
        for rep in range(2):
            torch.cuda.synchronize(); t = time.time()
            mean, sd = gd.reconstruct_slices(Rs, axis=-1, batch=batch, batch_concurrency=conc, **kw)
            torch.cuda.synchronize(); dt = time.time() - t
        print("%d slices: batch %d x concurrency %d: %.3f s" % (nsl, batch, conc, dt), flush=True)
