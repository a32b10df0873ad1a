"""Rank 0's share of config C3 at world size 8 (8 slices) under different (batch, concurrency) splits."""
import sys, os, time
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
sub = R[..., 0::8]
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
for batch, conc in ((1, 8), (1, 4), (2, 4), (4, 2), (8, 1), ("auto", None)):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        if batch == "auto": gd.reconstruct_slices(sub, axis=-1, batch="auto", **kw)
        else: gd.reconstruct_slices(sub, axis=-1, batch=batch, batch_concurrency=conc, **kw)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print("8 slices, batch %s x concurrency %s: %.3f s" % (batch, conc, best), flush=True)
