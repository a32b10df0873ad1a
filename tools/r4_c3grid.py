import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tests")
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
for batch, conc in ((16, 4), (8, 8), (8, 4), (22, 3), (11, 6), (4, 8), (4, 16)):
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        mean, sd = gd.reconstruct_slices(R, axis=-1, batch=batch, batch_concurrency=conc, **kw)
        torch.cuda.synchronize(); dt = time.time() - t
    print("64 slices: batch %d x concurrency %d: %.3f s" % (batch, conc, dt), flush=True)
