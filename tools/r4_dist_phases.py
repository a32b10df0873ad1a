"""Phase times of ONE Adam iteration of the distributed training loop (gpim_amd/dist_chol.py exact_gp_fit) on one GPU:
K build, factorisation, vector solves, streamed inverse, K^-1 pass, gradient sums.   usage: r4_dist_phases.py N"""
import sys, os, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
from gpim_amd.dist_chol import DistributedCholesky, PW
from problems import lattice_image
N = int(sys.argv[1])
side = int(round(np.sqrt(N))); N = side * side
R, _ = lattice_image(size=side, frac=1.0, seed=1)
ii, jj = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="ij")
X = torch.from_numpy(np.stack([ii.ravel(), jj.ravel()], 1)); y = torch.from_numpy(R.ravel().astype(np.float64))
spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5)
m = spec.struct()
chol = DistributedCholesky(N)
L, H = chol.layout, chol.engine.H
dev, lib = H.device, H.lib
Xd, yd = X.to(dev).contiguous(), y.to(dev).contiguous()
u = spec.draw_initial_u(torch.Generator().manual_seed(0)).to(dev).contiguous()
S = torch.zeros((8,), dtype=torch.float64, device=dev)
alpha_pad = torch.zeros((L.np,), dtype=torch.float64, device=dev)
ld = chol.local.stride(0)
def timed(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print("  %-14s %8.1f ms" % (name, dt * 1e3), flush=True)
    return r, dt
for rep in range(2):
    print("iteration", rep)
    tot = 0.0
    def build():
        for p in L.owned:
            _lib.check(lib.gpimhip_dist_kmat_cols(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), p * PW, L.width(p),
                                                  ctypes.c_void_p(chol.local.data_ptr() + 8 * L.local_col0(p)), ld))
    _, dt = timed("kmat", build); tot += dt
    _, dt = timed("factor", chol.factor); tot += dt
    alpha, dt = timed("solve", lambda: chol.solve(yd)); tot += dt
    alpha_pad[:N] = alpha
    Xl, dt = timed("inverse", chol.inverse); tot += dt
    Kl, dt = timed("kinv", lambda: chol.kinv(Xl)); tot += dt
    del Xl
    _, dt = timed("grad_sums", lambda: _lib.check(lib.gpimhip_dist_grad_sums(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), _lib.ptr(Kl), Kl.stride(0), _lib.ptr(alpha_pad), _lib.ptr(S)))); tot += dt
    del Kl
    f = float(N) ** 3 / 3 / 1e12
    print("  total %.1f ms; N^3/3 = %.2f TFLOP per O(N^3) pass" % (tot * 1e3, f))
