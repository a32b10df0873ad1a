import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
ref = None
for batch, conc in ((64, 1), (32, 2), (16, 4), (8, 4), (16, 2)):
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        mean, sd = gd.reconstruct_slices(R, axis=-1, batch=batch, batch_concurrency=conc, **kw)
        torch.cuda.synchronize(); dt = time.time() - t
    if ref is None: ref = mean
    print("C3 batch %d x concurrency %d: %.3f s -> %.0f grid-points/s, same bits %s" % (batch, conc, dt, 64 * 4096 / dt, np.array_equal(mean, ref)), flush=True)
