import sys; sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..")); sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
import numpy as np
import gpim_amd as gpim
from gpim_amd import dist as gd
from problems import ckpfm_cube
cube = ckpfm_cube()
for T in (25, 200):
    kw = dict(kernel="RBF", sparse=True, indpoints=512, learning_rate=0.05, iterations=T)
    mb, sb, hb = gd.reconstruct_slices(cube, axis=-1, return_hyperparams=True, **kw)
    m0, s0, h0 = gd.reconstruct_slices(cube, axis=-1, return_hyperparams=True, sparse_batch=0, **kw)
    print(T, "mean maxdiff %.3e sd maxdiff %.3e" % (np.abs(mb - m0).max(), np.abs(sb - s0).max()))
    for k in range(5):
        print("  slice", k, "noise rel %.3e ls rel %.3e xu max %.3e" % (
            np.max(np.abs(np.array(hb[k]["noise"]) / np.array(h0[k]["noise"]) - 1)),
            np.max(np.abs(np.array(hb[k]["lengthscale"]) / np.array(h0[k]["lengthscale"]) - 1)),
            np.abs(hb[k]["inducing_points"][-1] - h0[k]["inducing_points"][-1]).max()))
