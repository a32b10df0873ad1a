"""What ONE rank of a P-rank job computes per Adam iteration of the distributed exact GP, timed on one GPU without the other
ranks: the collectives are replaced by no-ops (the panels this rank does not own arrive as whatever the buffer holds --
the arithmetic runs on garbage at the same speed; no result of this script means anything).  Compute-only view of strong
scaling: phase times at world = 1, 2, 4, 8 for rank 0.      usage: r5_dist_rank_share.py N [worlds...]"""
import sys, os, time, ctypes, types
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from gpim_amd import _lib, dist_chol
from gpim_amd.kernels import KernelSpec
from gpim_amd.dist_chol import DistributedCholesky, Layout, PW
from problems import lattice_image

class _Done:
    def wait(self): return True
fake = types.SimpleNamespace(broadcast=lambda *a, **k: _Done(), all_reduce=lambda *a, **k: None, is_available=lambda: False,
                             is_initialized=lambda: False, ReduceOp=types.SimpleNamespace(MAX=0))
N = int(sys.argv[1]); worlds = [int(w) for w in sys.argv[2:]] or [1, 2, 4, 8]
side = int(round(np.sqrt(N))); N = side * side
R, _ = lattice_image(size=side, frac=1.0, seed=1)
ii, jj = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="ij")
X = torch.from_numpy(np.stack([ii.ravel(), jj.ravel()], 1)); y = torch.from_numpy(R.ravel().astype(np.float64))
spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5); m = spec.struct()
for world in worlds:
    dist_chol.dist = fake
    dist_chol._world = lambda w=world: (0, w)
    chol = DistributedCholesky(N)
    L, H = chol.layout, chol.engine.H
    dev, lib = H.device, H.lib
    Xd, yd = X.to(dev).contiguous(), y.to(dev).contiguous()
    u = spec.draw_initial_u(torch.Generator().manual_seed(0)).to(dev).contiguous()
    S = torch.zeros((10,), dtype=torch.float64, device=dev)
    alpha_pad = torch.zeros((L.np,), dtype=torch.float64, device=dev)
    ld = chol.local.stride(0)
    def timed(fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        return r, (time.perf_counter() - t) * 1e3
    def build():
        for p in L.owned:
            _lib.check(lib.gpimhip_dist_kmat_cols(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), p * PW, L.width(p),
                                                  ctypes.c_void_p(chol.local.data_ptr() + 8 * L.local_col0(p)), ld))
    for rep in range(2):
        _, t_k = timed(build)
        _, t_f = timed(lambda: chol.factor(check=False))
        alpha, t_s = timed(lambda: chol.solve(yd))
        Xl, t_i = timed(chol.inverse)
        Kl, t_ki = timed(lambda: chol.kinv(Xl, out=chol.local))
        _, t_g = timed(lambda: _lib.check(lib.gpimhip_dist_grad_sums(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), _lib.ptr(Kl), Kl.stride(0), _lib.ptr(alpha_pad), _lib.ptr(S))))
    tot = t_k + t_f + t_s + t_i + t_ki + t_g
    print("world %d rank 0 (%d of %d panels): kmat %.0f | factor %.0f | solve %.0f | inverse %.0f | kinv %.0f | grad %.0f | total %.0f ms"
          % (world, len(L.owned), L.npanel, t_k, t_f, t_s, t_i, t_ki, t_g, tot), flush=True)
    del chol, Xl, Kl
    torch.cuda.empty_cache()
