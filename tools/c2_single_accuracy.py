import sys, os, time; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gpim_amd as gpim
from problems import lattice_image
R2, truth = lattice_image(size=256, frac=0.25, seed=1)
X2, Xf2 = gpim.utils.get_sparse_grid(R2), gpim.utils.get_full_grid(R2)
res = {}
for prec in ("double", "single"):
    rec = gpim.reconstructor(X2, R2, Xf2, iterations=100, kernel="Matern52", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, verbose=0, seed=0, precision=prec)
    if prec == "single":
        rec._u.copy_(u0)
    else:
        u0 = rec._u.clone()
    m, s, hp = rec.run()
    res[prec] = (np.asarray(m, dtype=np.float64), np.asarray(s, dtype=np.float64), hp)
md, sd = res["double"][:2]; ms, ss = res["single"][:2]
print("C2 after 100 Adam iterations from the same start: mean max|single - double| %.2e (rmse %.2e), sd max rel %.2e; rmse vs truth double %.5f single %.5f" % (
    np.abs(ms - md).max(), np.sqrt(np.mean((ms - md) ** 2)), (np.abs(ss - sd) / sd).max(), np.sqrt(np.mean((md - truth) ** 2)), np.sqrt(np.mean((ms - truth) ** 2))))
for k in ("lengthscale", "noise", "variance"):
    if k in res["double"][2]:
        print(k, np.asarray(res["double"][2][k][-1]), np.asarray(res["single"][2][k][-1]))
