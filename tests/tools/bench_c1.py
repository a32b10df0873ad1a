"""Config C1 (synthetic twin): 128x128 spiral-masked image, N ~ 4206, RBF, T = 300, reconstructor.run()."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpim_amd as gpim
from problems import spiral_image
R, _ = spiral_image()
X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    mean, sd, hp = gpim.reconstructor(X, R, Xf, kernel="RBF", lengthscale=[[1., 1.], [4., 4.]], learning_rate=0.1,
                                      iterations=300, verbose=0).run()
    torch.cuda.synchronize(); dt = time.time() - t
    print("C1: N=%d M=%d T=300: %.2f s -> %.0f grid-points/s (%.2f ms per Adam iteration incl. predict)" % (
        np.isfinite(R).sum(), R.size, dt, R.size / dt, dt / 300 * 1e3))
