import sys, os, time
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
gd.reconstruct_slices(R, axis=-1, batch=16, batch_concurrency=4, **dict(kw, iterations=3))
for b, c in ((16, 4), (8, 8), (8, 4), (11, 6), (13, 5), (22, 3), (32, 2), (16, 4)):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        gd.reconstruct_slices(R, axis=-1, batch=b, batch_concurrency=c, **kw)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print("C3 batch %d x conc %d: %.3f s" % (b, c, best), flush=True)
