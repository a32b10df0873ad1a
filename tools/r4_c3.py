"""Config C3 on one GPU (64 slices of 64x64, N = 1207, T = 250) at several batch sizes x concurrencies, and the same
slices bitwise against stand-alone fits.   python tools/r4_c3.py [check]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
if len(sys.argv) > 1 and sys.argv[1] == "check":
    kw["iterations"] = 20
    ref = gd.reconstruct_slices(R[..., :6], axis=-1, batch=1, batch_concurrency=1, **kw)
    for batch in (6, 5, 3):
        got = gd.reconstruct_slices(R[..., :6], axis=-1, batch=batch, batch_concurrency=1, **kw)
        print("batch %d vs stand-alone: mean equal %s, sd equal %s, max|dmean| %.3e" % (
            batch, np.array_equal(ref[0], got[0]), np.array_equal(ref[1], got[1]), np.abs(ref[0] - got[0]).max()), flush=True)
    sys.exit(0)
for batch, conc in ((64, 1), (32, 2), (16, 4), (32, 1), (16, 1)):
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        mean, sd = gd.reconstruct_slices(R, axis=-1, batch=batch, batch_concurrency=conc, **kw)
        torch.cuda.synchronize(); dt = time.time() - t
    print("64 slices: batch %d x concurrency %d: %.3f s" % (batch, conc, dt), flush=True)
