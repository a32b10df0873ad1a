"""Config C3 on one GPU: the part that does not depend on the number of Adam iterations (host preparation, uploads, posterior,
gather) -- T = 1 against T = 250 -- and its split."""
import sys, os, time, cProfile, pstats
import numpy as np, torch
R0 = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
from gpim_amd import dist as gd
from problems import hyperspectral_cube
import gc
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, verbose=0)
gd.reconstruct_slices(R, axis=-1, batch=16, batch_concurrency=4, iterations=3, **kw)
gc.collect(); gc.freeze()
for T in (1, 250, 1, 250):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        gd.reconstruct_slices(R, axis=-1, batch=16, batch_concurrency=4, iterations=T, **kw)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("C3 16x4, T = %3d: %.3f s" % (T, best), flush=True)
pr = cProfile.Profile(); pr.enable()
gd.reconstruct_slices(R, axis=-1, batch=16, batch_concurrency=1, iterations=1, **kw)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
