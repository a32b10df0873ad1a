"""Config C3 and rank 0's share at world size 8 under (batch, concurrency) splits -- run with GPU_MAX_HW_QUEUES=4 (the
runtime's default) and =8 / 16: do more than four concurrent streams help once they get hardware queues of their own?"""
import sys, os, time
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
from gpim_amd import dist as gd
from problems import hyperspectral_cube
R, _ = hyperspectral_cube(size=64, nspec=64)
kw = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250, verbose=0)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for name, cube, splits in (("share (8 slices)", R[..., 0::8], ((4, 2), (2, 4), (1, 8))),
                           ("C3 (64 slices)", R, ((16, 4), (8, 8), (4, 16), (11, 6)))):
    for batch, conc in splits:
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize(); t = time.time()
            gd.reconstruct_slices(cube, axis=-1, batch=batch, batch_concurrency=conc, **kw)
            torch.cuda.synchronize(); best = min(best, time.time() - t)
        print("%s, batch %s x concurrency %s: %.3f s" % (name, batch, conc, best), flush=True)
