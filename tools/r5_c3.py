"""Config C3 on one GPU (64 slices, 16 x 4 concurrent) and rank 0's share at world size 8 (8 slices, batch='auto')."""
import sys, os, time
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
for name, cube, bkw in (("C3 16x4", R, dict(batch=16, batch_concurrency=4)), ("C3 batch 64", R, dict(batch=64)),
                        ("per-rank-8 auto", R[..., 0::8], dict(batch="auto"))):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        gd.reconstruct_slices(cube, axis=-1, **bkw, **kw)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print("%s: %.3f s" % (name, best), flush=True)
