"""Config C3 on one GPU: 64 independent 64x64 slices (N~1200 each), T iterations, by concurrency."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpim_amd
from gpim_amd import dist as gd
from problems import hyperspectral_cube
T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
nsl = int(sys.argv[2]) if len(sys.argv) > 2 else 16
R, _ = hyperspectral_cube(size=64, nspec=nsl)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=T, verbose=0)
ref = None
for conc in (1, 2, 4, 8, 16, 64):
    torch.cuda.synchronize(); t = time.time()
    mean, sd = gd.reconstruct_slices(R, axis=-1, batch=conc, **kw)
    torch.cuda.synchronize(); dt = time.time() - t
    if ref is None: ref = (mean, sd)
    print(f"batch {conc:2d}: {dt:.2f} s for {nsl} slices x {T} its -> {nsl*4096/dt:.0f} grid-points/s; "
          f"max|dmean| vs serial {np.abs(mean-ref[0]).max():.2e}")
