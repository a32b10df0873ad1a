"""The streamed inverse of the distributed model alone (P = 1), bracketed by marker kernels (cumsum) for a kernel trace.
usage: r5_dist_inv_trace.py N"""
import sys, os, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
from gpim_amd.dist_chol import DistributedCholesky, PW
from problems import lattice_image
N = int(sys.argv[1]); side = int(round(np.sqrt(N))); N = side * side
R, _ = lattice_image(size=side, frac=1.0, seed=1)
ii, jj = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="ij")
X = torch.from_numpy(np.stack([ii.ravel(), jj.ravel()], 1))
spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5); m = spec.struct()
chol = DistributedCholesky(N); L, H = chol.layout, chol.engine.H; dev, lib = H.device, H.lib
Xd = X.to(dev).contiguous(); u = spec.draw_initial_u(torch.Generator().manual_seed(0)).to(dev).contiguous()
ld = chol.local.stride(0)
for p in L.owned:
    _lib.check(lib.gpimhip_dist_kmat_cols(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), p * PW, L.width(p), ctypes.c_void_p(chol.local.data_ptr() + 8 * L.local_col0(p)), ld))
chol.factor()
chol.inverse()
mark = torch.arange(7, device=dev, dtype=torch.float64)
torch.cuda.synchronize(); mark.cumsum(0); t = time.perf_counter()
chol.inverse()
mark.cumsum(0); torch.cuda.synchronize()
print("inverse %.1f ms" % ((time.perf_counter() - t) * 1e3))
