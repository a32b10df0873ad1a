"""Phase times of one Adam iteration of the distributed exact GP PER RANK (launch under torchrun, one rank per GPU):
K build, factorisation, vector solves, streamed inverse, K^-1 pass, gradient sums; every phase bracketed by a device
synchronisation and a barrier, so a rank's number includes what it waited for.   usage: r5_dist_phases_mp.py N"""
import sys, os, time, ctypes
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpim_amd import _lib
from gpim_amd import dist as gd
from gpim_amd.kernels import KernelSpec
from gpim_amd.dist_chol import DistributedCholesky, PW
from problems import lattice_image
rank, world, _ = gd.init_from_env()
N = int(sys.argv[1]); side = int(round(np.sqrt(N))); N = side * side
R, _ = lattice_image(size=side, frac=1.0, seed=1)
ii, jj = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="ij")
X = torch.from_numpy(np.stack([ii.ravel(), jj.ravel()], 1)); y = torch.from_numpy(R.ravel().astype(np.float64))
spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5); m = spec.struct()
chol = DistributedCholesky(N)
L, H = chol.layout, chol.engine.H
dev, lib = H.device, H.lib
Xd, yd = X.to(dev).contiguous(), y.to(dev).contiguous()
u = spec.draw_initial_u(torch.Generator().manual_seed(0)).to(dev).contiguous()
S = torch.zeros((10,), dtype=torch.float64, device=dev)
alpha_pad = torch.zeros((L.np,), dtype=torch.float64, device=dev)
ld = chol.local.stride(0)
def timed(fn):
    torch.cuda.synchronize()
    if world > 1: dist.barrier()
    t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t) * 1e3
def build():
    for p in L.owned:
        _lib.check(lib.gpimhip_dist_kmat_cols(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), p * PW, L.width(p),
                                              ctypes.c_void_p(chol.local.data_ptr() + 8 * L.local_col0(p)), ld))
for rep in range(2):
    _, t_k = timed(build)
    _, t_f = timed(chol.factor)
    alpha, t_s = timed(lambda: chol.solve(yd))
    alpha_pad[:N] = alpha
    Xl, t_i = timed(chol.inverse)
    Kl, t_ki = timed(lambda: chol.kinv(Xl, out=chol.local))
    _, t_g = timed(lambda: _lib.check(lib.gpimhip_dist_grad_sums(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), _lib.ptr(Kl), Kl.stride(0), _lib.ptr(alpha_pad), _lib.ptr(S))))
    line = "iteration %d rank %d/%d: kmat %.0f | factor %.0f | solve %.0f | inverse %.0f | kinv %.0f | grad %.0f | total %.0f ms" % (
        rep, rank, world, t_k, t_f, t_s, t_i, t_ki, t_g, t_k + t_f + t_s + t_i + t_ki + t_g)
    for r in range(world):
        if r == rank: print(line, flush=True)
        if world > 1: dist.barrier()
if world > 1: dist.destroy_process_group()
