"""Diagnostics for config C5 at full size: how far the HIP sparse-VFE run and the oracle drift apart over
25 Adam iterations, per quantity, and how that relates to the size of the gradient components (Adam
normalises every coordinate's step to ~lr whatever the gradient's magnitude, so coordinates whose
gradient is rounding noise random-walk)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpim_amd as gpim
from oracle import gpim_oracle as O
from problems import ckpfm_cube

T = int(sys.argv[1]) if len(sys.argv) > 1 else 25
cube = ckpfm_cube()
R = cube[..., 1]
Xf = gpim.utils.get_full_grid(R)
kw = dict(kernel="RBF", sparse=True, indpoints=512, learning_rate=0.05, iterations=T, verbose=0)
rec = gpim.reconstructor(Xf, R, Xf, **kw)
m, s, h = rec.run()
torch.set_num_threads(32)
orc = O.reconstructor(Xf, R, Xf, **kw)
params = orc.model.parameters()
opt = torch.optim.Adam(params, lr=0.05)
gmin = None
hist_xu = []
for it in range(T):
    opt.zero_grad()
    loss = orc.model.loss()
    loss.backward()
    g = orc.model.Xu.grad.abs().clone()
    gmin = g if gmin is None else torch.minimum(gmin, g)
    opt.step()
    hist_xu.append(orc.model.Xu.detach().numpy().copy())
    d = np.abs(h["inducing_points"][it] - hist_xu[-1])
    print("it %2d  max|dXu| %.3e  noise rel %.2e  ls rel %.2e  |g| quantiles %s" % (
        it, d.max(), abs(h["noise"][it] - orc.kernel.noise.item()) / orc.kernel.noise.item(),
        np.max(np.abs(np.array(h["lengthscale"][it]) - np.array(orc.kernel.lengthscale.tolist())) /
               np.array(orc.kernel.lengthscale.tolist())),
        np.quantile(g.numpy(), [0.0, 0.01, 0.1, 0.5]).round(10)))
mo, so = orc.predict()
d = np.abs(h["inducing_points"][-1] - hist_xu[-1])
gm = gmin.numpy()
for thr in (1e-8, 1e-7, 1e-6, 1e-5, 1e-4):
    well = gm > thr
    print("min|g| > %.0e: %4d of %d coords, max |dXu| there %.3e" % (thr, well.sum(), well.size, d[well].max() if well.any() else 0))
print("mean max abs diff %.3e, sd max abs diff %.3e" % (np.abs(m - mo).max(), np.abs(s - so).max()))
