"""
dist_symm.py -- ONE exact GP on a complete grid across the GPUs of a node WITHOUT a data-path collective: the reflection
blocks of the symmetry-reduced model (gprutils.reflection_blocks; csrc/engine.hip: kmat_refl_kernel) are dealt to the ranks.

The reference fits whole images / cubes as one GP (gpim/gpreg/gpr.py:30-43,115-126).  On a complete grid the covariance of a
stationary kernel that is even in every coordinate difference is block diagonal in the reflection-adapted basis: 2^r dense
blocks (4 for an image, 8 for a cube) that share nothing but the hyper-parameters.  Block s goes to rank s mod P; per Adam
iteration every rank evaluates its blocks (gpimhip_refl_sums: kernel matrices, factorisations, inverses, gradient
contraction), ONE all-reduce of eleven doubles adds the gradient sums, log-determinants and quadratic forms, and the chain
rule + Adam step of the single-GPU path runs replicated (gpimhip_dist_finalize_dev).  The 64 x 64 x 64 cube of config C3
(N = 262144): one block of 32768 points per GPU on eight GPUs, 24 GiB each.  The prediction adds the blocks' shares of the
mean and of the variance's quadratic form with one all-reduce of 2 M doubles.

world = 1 reproduces ``reconstructor(structured=True)`` (same kernels, the sums taken in the same order).
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist

from . import _lib, gprutils
from .kernels import KernelSpec

_F64 = torch.float64


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _grid_axes(X):
    from .gpr import reconstructor
    return reconstructor._grid_axes(X)[0]


class _Shard:
    """This rank's blocks on its GPU: the handle in reflection mode, the fundamental domain, the projected observations."""

    def __init__(self, X, y, spec, rank, world):
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        if np.isnan(y).any():
            raise NotImplementedError("the symmetry-reduced model needs a fully observed grid (no NaN in y)")
        S = gprutils.reflection_blocks(X, y, _grid_axes(X))
        self.S, self.B = S, S["B"]
        self.mine = list(range(rank, self.B, world))
        self.H = _lib.Handle()
        self.dev = self.H.device
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev, _F64).contiguous()
        self.Xq = to(S["Xq"])
        self.Nq = self.Xq.shape[0]
        self.n_total = S["n_total"]
        self.twoc = (ctypes.c_double * 4)(*S["twoc"])
        if self.mine:
            self.ys = to(S["ys"][self.mine])
            self.wts = to(S["wts"][self.mine]) if S["wts"] is not None else None
        self.m = spec.struct()
        self.P = spec.n_params
        self.rank, self.world = rank, world

    def enter(self, raw=0, var_count=0):
        lib, h = self.H.lib, self.H.h
        _lib.check(lib.gpimhip_set_reflection(h, self.S["mask"], self.twoc, _lib.ptr(self.wts) if self.mine else None,
                                              self.n_total, int(var_count)))
        _lib.check(lib.gpimhip_set_reflection_shard(h, self.rank, self.world, self.B, int(raw)))

    def leave(self):
        _lib.check(self.H.lib.gpimhip_set_reflection(self.H.h, 0, None, None, 0, 0))


def symm_gp_fit(X, y, kernel="Matern52", lengthscale=None, learning_rate=5e-2, iterations=100, seed=0, jitter=1e-5,
                amplitude=None, group=None, verbose=0, u0=None, shard=None):
    """Trains ONE exact GP on the complete grid X (d, n_1, ..., n_d) / y (n_1, ..., n_d) across the ranks of the process
    group: the training loop of ``reconstructor.train`` (gpim/gpreg/gpr.py:170-217).  Every rank passes the same arguments
    and ends with the same hyper-parameters.  Returns (hyperparams, u) like ``dist_chol.exact_gp_fit``.
    shard: a ``symm_shard(X, y, ...)`` of the same model, to share the rank's blocks and workspace with the posterior."""
    rank, world = _world()
    y = np.asarray(y, dtype=np.float64)
    d = y.ndim
    if lengthscale is None:
        lengthscale = [[0.0] * d, [float(np.mean(y.shape) / 2)] * d]
    spec = KernelSpec(kernel, d, lengthscale, amplitude=amplitude, jitter=jitter)
    sh = shard if shard is not None else _Shard(X, y, spec, rank, world)
    lib, H, dev, P = sh.H.lib, sh.H, sh.dev, sh.P
    u = (spec.draw_initial_u(torch.Generator().manual_seed(seed)) if u0 is None
         else torch.as_tensor(u0, dtype=_F64).clone()).to(dev).contiguous()
    T = int(iterations)
    hist = torch.zeros((max(T, 1), P), dtype=_F64, device=dev)
    loss = torch.zeros((max(T, 1),), dtype=_F64, device=dev)
    sums = torch.zeros((11,), dtype=_F64, device=dev)
    back = torch.zeros((2,), dtype=_F64).pin_memory()
    sh.enter()
    try:
        for t in range(1, T + 1):
            if sh.mine:
                u_b = u.repeat(len(sh.mine)).contiguous()
                _lib.check(lib.gpimhip_refl_sums(H.h, ctypes.byref(sh.m), _lib.ptr(sh.Xq), _lib.ptr(sh.ys), sh.Nq, len(sh.mine),
                                                 _lib.ptr(u_b), _lib.ptr(sums)))
            else:
                sums.zero_()
            if world > 1:
                dist.all_reduce(sums, group=group)
            _lib.check(lib.gpimhip_dist_finalize_dev(H.h, ctypes.byref(sh.m), sh.n_total, _lib.ptr(u), _lib.ptr(sums),
                                                     ctypes.c_void_p(sums.data_ptr() + 80), float(learning_rate), t,
                                                     ctypes.c_void_p(loss[t - 1:].data_ptr()), None,
                                                     ctypes.c_void_p(hist[t - 1].data_ptr())))
            back[0:1].copy_(sums[9:10], non_blocking=True)
            back[1:2].copy_(loss[t - 1:t], non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()
            if back[0].item() != 0:
                raise torch.linalg.LinAlgError("linalg.cholesky: the input is not positive-definite")
            if verbose and rank == 0 and (t == 1 or t % 10 == 0 or t == T):
                print("iter: {} ... loss: {:.4f}".format(t - 1, float(back[1].item())))
    finally:
        sh.leave()
    hcpu = hist[:T].cpu().numpy()
    hyper = {"variance": hcpu[:, 0], "lengthscale": hcpu[:, 1:1 + spec.n_ls], "noise": hcpu[:, 1 + spec.n_ls],
             "loss": loss[:T].cpu().numpy()}
    return hyper, u


def symm_shard(X, y, kernel="Matern52", lengthscale=None, jitter=1e-5, amplitude=None):
    """This rank's share of the model (its reflection blocks on its GPU, one library handle): pass it to ``symm_gp_fit`` and
    ``symm_gp_posterior`` as ``shard=`` so that both use the same workspace (24 GiB per block of the 64^3 cube)."""
    rank, world = _world()
    y = np.asarray(y, dtype=np.float64)
    d = y.ndim
    if lengthscale is None:
        lengthscale = [[0.0] * d, [float(np.mean(y.shape) / 2)] * d]
    return _Shard(X, y, KernelSpec(kernel, d, lengthscale, amplitude=amplitude, jitter=jitter), rank, world)


def symm_gp_posterior(X, y, Xtest, u, kernel="Matern52", lengthscale=None, jitter=1e-5, amplitude=None, group=None, shard=None):
    """Posterior mean and standard deviation (noise included) of the same model at the points Xtest (M, d) for the
    unconstrained parameters u (as returned by ``symm_gp_fit``): every rank adds its blocks' shares, one all-reduce of 2 M
    doubles.  Xtest=None predicts on the training grid itself (row-major order of y): the variance is invariant under the
    reflections, so its quadratic form is computed on the fundamental domain only (an eighth of a cube) and mirrored, the
    way ``reconstructor(structured=True).predict`` does it.  Returns (mean, sd) numpy vectors, the same on every rank."""
    rank, world = _world()
    y = np.asarray(y, dtype=np.float64)
    d = y.ndim
    if lengthscale is None:
        lengthscale = [[0.0] * d, [float(np.mean(y.shape) / 2)] * d]
    spec = KernelSpec(kernel, d, lengthscale, amplitude=amplitude, jitter=jitter)
    sh = shard if shard is not None else _Shard(X, y, spec, rank, world)
    lib, H, dev = sh.H.lib, sh.H, sh.dev
    perm = None
    if Xtest is None:
        S = sh.S
        pts = torch.as_tensor(np.asarray(X, dtype=np.float64).reshape(d, -1).T.copy()).to(dev)
        rest = np.setdiff1d(np.arange(pts.shape[0]), S["fund_flat"], assume_unique=True)
        perm = torch.from_numpy(np.concatenate([S["fund_flat"], rest])).to(dev)
        rep = torch.from_numpy(S["rep"]).to(dev)
        Xt, nq = pts[perm].contiguous(), len(S["fund_flat"])
    else:
        Xt, nq = torch.as_tensor(np.asarray(Xtest, dtype=np.float64)).to(dev).contiguous(), 0
    M = Xt.shape[0]
    ud = torch.as_tensor(u, dtype=_F64).to(dev).contiguous()
    both = torch.zeros((2, M), dtype=_F64, device=dev)
    if sh.mine:
        sh.enter(raw=1, var_count=nq)
        try:
            u_b = ud.repeat(len(sh.mine)).contiguous()
            _lib.check(lib.gpimhip_predict_exact_batched(H.h, ctypes.byref(sh.m), _lib.ptr(sh.Xq), 0, _lib.ptr(sh.ys), sh.Nq,
                                                         len(sh.mine), _lib.ptr(u_b), _lib.ptr(Xt), M,
                                                         ctypes.c_void_p(both[0].data_ptr()), ctypes.c_void_p(both[1].data_ptr())))
        finally:
            sh.leave()
    if world > 1:
        dist.all_reduce(both, group=group)
    var, _, noise = spec.constrained(ud)
    mean, q = both[0], both[1]
    if perm is not None:
        mean = torch.empty_like(both[0])
        mean[perm] = both[0]
        q = both[1][:nq][rep]
    sd = torch.sqrt(torch.clamp(float(var) - q, min=0.0) + float(noise))
    return mean.cpu().numpy(), sd.cpu().numpy()
