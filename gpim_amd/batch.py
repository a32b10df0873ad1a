"""
batch.py -- lock-step fitting of many independent exact GPs of equal size on one GPU.

The reference fits one GP per ``reconstructor(...).run()`` call (gpr.py:257-283) and has no slice
loop; a hyperspectral cube reconstructed slice by slice (config C3 of SURVEY 8(d)) is a Python loop
over such calls.  A ~1000-point fit is latency-bound on an MI355X (the blocked Cholesky's panel chain
leaves most of the chip idle), so here B slices advance together through every launch
(gpimhip_fit_exact_batched / gpimhip_predict_exact_batched: grid.y = slice).  Each slice still gets
exactly the arithmetic of its own ``reconstructor``: same seed -> same initial draw, same Adam loop,
same outputs.
"""
import ctypes

import numpy as np
import torch

from . import _lib, gprutils
from .kernels import get_kernel

_F64 = torch.float64


def fit_predict_batch(Xs_list, ys_list, Xtest, kernel='RBF', lengthscale=None, learning_rate=5e-2,
                      iterations=1000, seed=0, handle=None, **kwargs):
    """Fit B exact GPs (problem b: sparse grid Xs_list[b] (c,*dims), observations ys_list[b] (*dims))
    and predict each on the shared grid Xtest.  All problems must have the same number of
    observations.  Returns (mean, sd, hist): arrays (B, *Xtest.shape[1:]) and the hyper-parameter
    history (B, iterations, P) in the order [variance, lengthscale.., noise(, alpha)]."""
    H = handle or _lib.Handle(precision=kwargs.get("precision", "double"))
    try:
        return _fit_predict_batch(H, Xs_list, ys_list, Xtest, kernel, lengthscale, learning_rate, iterations, seed, **kwargs)
    finally:
        if handle is None:
            H.close()


def _fit_predict_batch(H, Xs_list, ys_list, Xtest, kernel, lengthscale, learning_rate, iterations, seed, **kwargs):
    dev = H.device
    B = len(ys_list)
    y0 = np.asarray(ys_list[0])
    input_dim = np.ndim(y0)
    if lengthscale is None and not kwargs.get("isotropic"):
        lmean = float(np.mean(y0.shape) / 2)
        lengthscale = [[0. for _ in range(input_dim)], [lmean for _ in range(input_dim)]]
    elif lengthscale is None:
        lengthscale = [0., float(np.mean(y0.shape) / 2)]
    spec = get_kernel(kernel, input_dim, lengthscale, amplitude=kwargs.get('amplitude'),
                      jitter=kwargs.get("jitter", 1.0e-5))
    m = spec.struct()
    P = spec.n_params
    shared_x = not isinstance(Xs_list, (list, tuple))
    Xl, yl = [], []
    for b in range(B):
        Xb, yb = gprutils.prepare_training_data(Xs_list if shared_x else Xs_list[b], ys_list[b])
        Xl.append(Xb)
        yl.append(yb)
    N = yl[0].shape[0]
    if any(t.shape[0] != N for t in yl) or any(t.shape[0] != N for t in Xl):
        raise ValueError("fit_predict_batch: every problem of a batch must have the same number of observations")
    same_x = all(torch.equal(Xl[0], t) for t in Xl[1:])
    Xd = (Xl[0] if same_x else torch.stack(Xl)).to(dev).contiguous()
    x_stride = 0 if same_x else N * input_dim
    yd = torch.stack(yl).to(dev).contiguous()
    # every slice is its own reconstructor(seed=seed): identical initial draw
    u0 = spec.draw_initial_u(torch.Generator().manual_seed(seed))
    u = u0.repeat(B, 1).to(dev).contiguous()
    T = int(iterations)
    hist = torch.empty((B, max(T, 1), P), dtype=_F64, device=dev)
    lib = H.lib
    _lib.check(lib.gpimhip_fit_exact_batched(H.h, ctypes.byref(m), _lib.ptr(Xd), x_stride, _lib.ptr(yd), N, B,
                                             _lib.ptr(u), float(learning_rate), T, _lib.ptr(hist), None))
    Xt = gprutils.prepare_test_data(Xtest).to(dev).contiguous()
    M = Xt.shape[0]
    mean = torch.empty((B, M), dtype=_F64, device=dev)
    var = torch.empty((B, M), dtype=_F64, device=dev)
    _lib.check(lib.gpimhip_predict_exact_batched(H.h, ctypes.byref(m), _lib.ptr(Xd), x_stride, _lib.ptr(yd), N, B,
                                                 _lib.ptr(u), _lib.ptr(Xt), M, _lib.ptr(mean), _lib.ptr(var)))
    shape = (B,) + tuple(Xtest.shape[1:])
    return mean.reshape(shape), var.sqrt().reshape(shape), hist[:, :T]


def fit_predict_batch_sparse(Xs_list, ys_list, Xtest, indpoints=None, kernel='RBF', lengthscale=None, learning_rate=5e-2,
                             iterations=1000, seed=0, handle=None, **kwargs):
    """B sparse (inducing-point, VFE) GPs of equal size in lock-step: what ``reconstructor(X_b, y_b, Xtest, sparse=True,
    indpoints=indpoints, ...).run()`` computes for every b (gpim/gpreg/gpr.py:145-155, once per slice of a 4D cube), all
    models in every launch (gpimhip_fit_vfe_batched / gpimhip_predict_vfe_batched).  Every problem needs the same number
    of observations (hence the same number of inducing inputs, X[::N // indpoints]).
    Returns (mean, sd, hist, hist_xu): (B, *Xtest.shape[1:]) twice, the hyper-parameter history (B, T, P) and the
    inducing inputs after every iteration (B, T, Mu, d)."""
    H = handle or _lib.Handle()
    try:
        return _fit_predict_batch_sparse(H, Xs_list, ys_list, Xtest, indpoints, kernel, lengthscale, learning_rate, iterations,
                                         seed, **kwargs)
    finally:
        if handle is None:
            H.close()        # the workspaces of B models go now, not when the collector gets to the object


def _fit_predict_batch_sparse(H, Xs_list, ys_list, Xtest, indpoints, kernel, lengthscale, learning_rate, iterations, seed,
                              **kwargs):
    dev = H.device
    B = len(ys_list)
    y0 = np.asarray(ys_list[0])
    input_dim = np.ndim(y0)
    if lengthscale is None and not kwargs.get("isotropic"):
        lmean = float(np.mean(y0.shape) / 2)
        lengthscale = [[0. for _ in range(input_dim)], [lmean for _ in range(input_dim)]]
    elif lengthscale is None:
        lengthscale = [0., float(np.mean(y0.shape) / 2)]
    spec = get_kernel(kernel, input_dim, lengthscale, amplitude=kwargs.get('amplitude'),
                      jitter=kwargs.get("jitter", 1.0e-5))
    m = spec.struct()
    P = spec.n_params
    shared_x = not isinstance(Xs_list, (list, tuple))
    Xl, yl = [], []
    for b in range(B):
        Xb, yb = gprutils.prepare_training_data(Xs_list if shared_x else Xs_list[b], ys_list[b])
        Xl.append(Xb)
        yl.append(yb)
    N = yl[0].shape[0]
    if any(t.shape[0] != N for t in yl) or any(t.shape[0] != N for t in Xl):
        raise ValueError("fit_predict_batch_sparse: every problem of a batch must have the same number of observations")
    # inducing inputs: every (N // indpoints)-th observation (reconstructor.__init__, gpim/gpreg/gpr.py:145-153)
    if indpoints is None:
        indpoints = N // 10
        indpoints = indpoints + 1 if indpoints == 0 else indpoints
    else:
        indpoints = N if indpoints > N else indpoints
    same_x = all(torch.equal(Xl[0], t) for t in Xl[1:])
    Xd = (Xl[0] if same_x else torch.stack(Xl)).to(dev, _F64).contiguous()
    x_stride = 0 if same_x else N * input_dim
    yd = torch.stack(yl).to(dev, _F64).contiguous()
    u0 = spec.draw_initial_u(torch.Generator().manual_seed(seed))          # every slice: its own reconstructor(seed=seed)
    Mu = len(Xl[0][::N // indpoints])
    u = torch.stack([torch.cat([u0, Xl[b][::N // indpoints].reshape(-1).to(_F64)]) for b in range(B)]).to(dev).contiguous()
    T = int(iterations)
    hist = torch.empty((B, max(T, 1), P), dtype=_F64, device=dev)
    hist_xu = torch.empty((B, max(T, 1), Mu, input_dim), dtype=_F64, device=dev)
    lib = H.lib
    _lib.check(lib.gpimhip_fit_vfe_batched(H.h, ctypes.byref(m), _lib.ptr(Xd), x_stride, _lib.ptr(yd), N, Mu, B, _lib.ptr(u),
                                           float(learning_rate), T, _lib.ptr(hist), _lib.ptr(hist_xu), None))
    Xt = gprutils.prepare_test_data(Xtest).to(dev, _F64).contiguous()
    M = Xt.shape[0]
    mean = torch.empty((B, M), dtype=_F64, device=dev)
    var = torch.empty((B, M), dtype=_F64, device=dev)
    _lib.check(lib.gpimhip_predict_vfe_batched(H.h, ctypes.byref(m), _lib.ptr(Xd), x_stride, _lib.ptr(yd), N, Mu, B, _lib.ptr(u),
                                               _lib.ptr(Xt), 0, M, _lib.ptr(mean), _lib.ptr(var)))
    shape = (B,) + tuple(Xtest.shape[1:])
    return mean.reshape(shape), var.sqrt().reshape(shape), hist[:, :T], hist_xu[:, :T]
