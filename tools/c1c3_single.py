import sys, os, time; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gpim_amd as gpim
from gpim_amd import dist as gdist
from problems import spiral_image, hyperspectral_cube
sync = torch.cuda.synchronize
R, _ = spiral_image(); X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
C1 = dict(kernel="RBF", lengthscale=[[1., 1.], [4., 4.]], learning_rate=0.1, iterations=300)
cube, _ = hyperspectral_cube()
C3 = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=250)
for prec in ("double", "single"):
    gpim.reconstructor(X, R, Xf, verbose=0, precision=prec, **dict(C1, iterations=3)).run()
    sync(); t0 = time.perf_counter(); gpim.reconstructor(X, R, Xf, verbose=0, precision=prec, **C1).run(); sync(); dt = time.perf_counter() - t0
    print("C1 %s: %.3f s (%.2f ms per iteration)" % (prec, dt, dt / 300 * 1e3), flush=True)
    gdist.reconstruct_slices(cube, axis=-1, batch=64, precision=prec, **dict(C3, iterations=3))
    sync(); t0 = time.perf_counter(); gdist.reconstruct_slices(cube, axis=-1, batch=64, precision=prec, **C3); sync(); dt = time.perf_counter() - t0
    print("C3 %s: %.3f s (%.2f ms per iteration of 64 problems)" % (prec, dt, dt / 250 * 1e3), flush=True)
