import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from gpim_amd import _lib
import test_gpu_single as T
from oracle import gpim_oracle as O
H32 = _lib.Handle(precision="single")
for kind, N, d in [("RBF", 300, 2), ("Matern52", 700, 2), ("RationalQuadratic", 260, 3), ("Matern52", 1500, 2), ("RBF", 2300, 2), ("RBF", 1000, 2)]:
    X, y, kp, spec, u, Xs = T.problem(N, d, kind, seed=N)
    gp = O.ExactGP(torch.from_numpy(X), torch.from_numpy(y), kp, 1e-5)
    lt, gt = gp.loss_and_grad(); lt = lt.item(); gt = gt.numpy()
    mt, vt = (t.numpy() for t in gp.predict(torch.from_numpy(Xs)))
    l, g, m, v = T.run_engine(_lib, H32, X, y, spec, u, Xs)
    lf, mf, vf = T.torch_float32_run(kp, X, y, Xs, 1e-5)
    K = kp.K(torch.from_numpy(X)).detach(); K.diagonal().add_(1e-5 + float(kp.noise))
    cond = torch.linalg.cond(K).item()
    print("%-18s N=%4d cond %.1e | loss err eng %.2e torch32 %.2e (rel %.1e) | grad rel %.1e | mean err eng %.2e t32 %.2e | var err eng %.2e t32 %.2e" % (
        kind, N, cond, abs(l - lt), abs(lf - lt), abs(l - lt) / abs(lt), np.abs(g - gt).max() / np.abs(gt).max(),
        np.abs(m - mt).max(), np.abs(mf - mt).max(), np.abs(v - vt).max(), np.abs(vf - vt).max()))
