"""Sparse-VFE timings on synthetic twins of the two runs whose wall-clock the reference publishes
(BASELINE.md section 1): 128x128 spiral with ~1053 inducing points (RBF, 300 its) and a 32x32x102 cube with
70% of the xy columns removed and ~1024 inducing points (Matern52, 500 its)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpim_amd
from problems import spiral_image
its = int(sys.argv[1]) if len(sys.argv) > 1 else 0
R, _ = spiral_image()
X, Xf = gpim_amd.utils.get_sparse_grid(R), gpim_amd.utils.get_full_grid(R)
for rep in range(2):
    t = time.time()
    rec = gpim_amd.reconstructor(X, R, Xf, kernel="RBF", lengthscale=[[1., 1.], [4., 4.]], sparse=True, indpoints=1000,
                                 learning_rate=0.1, iterations=its or 300, verbose=0)
    rec.train(); torch.cuda.synchronize(); t1 = time.time()
    rec.predict(); torch.cuda.synchronize(); t2 = time.time()
print("spiral 128x128 N=%d Mu=%d T=%d: train %.2f s (%.2f ms/iter), predict %.3f s -> %.0f grid-points/s (reference: 65.6-73.7 ms/iter, 17.7-19.1 s training, Colab GPU)"
      % (rec.X.shape[0], rec._n_ind, rec.iterations, t1 - t, (t1 - t) / rec.iterations * 1e3, t2 - t1, 16384 / (t2 - t)))
rng = np.random.default_rng(0)
ii, jj, ll = np.meshgrid(np.arange(32), np.arange(32), np.arange(102), indexing="ij")
cube = (1 + 0.5 * np.sin(ii / 5.0) * np.cos(jj / 4.0)) / (1 + ((ll - 51 - 10 * np.sin((ii + jj) / 9.0)) / 10.0) ** 2) + 0.01 * rng.standard_normal(ii.shape)
mask = rng.random((32, 32)) < 0.7
cube[mask] = np.nan
X, Xf = gpim_amd.utils.get_sparse_grid(cube), gpim_amd.utils.get_full_grid(cube)
for rep in range(2):
    t = time.time()
    rec = gpim_amd.reconstructor(X, cube, Xf, kernel="Matern52", lengthscale=[[1., 1., 1.], [20., 20., 20.]], sparse=True,
                                 indpoints=1000, learning_rate=0.1, iterations=its or 500, verbose=0)
    rec.train(); torch.cuda.synchronize(); t1 = time.time()
    rec.predict(); torch.cuda.synchronize(); t2 = time.time()
print("cube 32x32x102 N=%d Mu=%d T=%d: train %.2f s (%.2f ms/iter), predict %.3f s -> %.0f grid-points/s (reference: 199.6-204.1 ms/iter, 98.5-101.1 s training, Tesla P100)"
      % (rec.X.shape[0], rec._n_ind, rec.iterations, t1 - t, (t1 - t) / rec.iterations * 1e3, t2 - t1, cube.size / (t2 - t)))
