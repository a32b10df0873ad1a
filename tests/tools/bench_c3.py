import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    mean, sd = gd.reconstruct_slices(R, axis=-1, batch=64, **kw)
    torch.cuda.synchronize(); dt = time.time() - t
    print("C3: 64 slices x 250 its, batch 64: %.2f s -> %.0f grid-points/s" % (dt, 64 * 4096 / dt))
