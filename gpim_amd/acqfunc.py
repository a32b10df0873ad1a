"""
acqfunc.py -- acquisition functions over the dense grid, evaluated on the GPU.

Mirror of the reference's gpim/gpbayes/acqfunc.py:11-92: same call
signatures and return values ``(acq, (mean, sd))`` as numpy arrays.  The element-wise sweep
(CB / EI / POI, ``Phi`` through erfc) and the incumbent ``nanmax`` run in libgpimhip
(gpimhip_acq, gpimhip_nanmax).  ``gpmodel`` is normally a ``gpim_amd.reconstructor``; any
object with ``predict(X, verbose=0) -> (mean, sd)`` works (its arrays are uploaded).

Reference quirk kept on purpose (the golden test_poi.npy depends on it): in
``probability_of_improvement`` the incumbent is the nanmax over BOTH the posterior means and
the posterior standard deviations at the observed points (acqfunc.py:86-88).
"""
import ctypes

import numpy as np
import torch

from . import _lib

_F64 = torch.float64
_handles = {}


def _handle_for(gpmodel):
    h = getattr(gpmodel, "_handle", None)
    if h is not None:
        return h
    dev = _lib.require_gpu()
    if dev.index not in _handles:
        _handles[dev.index] = _lib.Handle()
    return _handles[dev.index]


def _device_pred(gpmodel, X, handle):
    """predict -> numpy results plus device copies (reused from the reconstructor if it has them)."""
    if hasattr(gpmodel, "_last_pred"):
        gpmodel._last_pred = None
    mean, sd = gpmodel.predict(X, verbose=0)
    last = getattr(gpmodel, "_last_pred", None)
    if last is not None:
        return mean, sd, last[0], last[1]
    md = torch.as_tensor(np.ascontiguousarray(mean), dtype=_F64).reshape(-1).to(handle.device)
    sdd = torch.as_tensor(np.ascontiguousarray(sd), dtype=_F64).reshape(-1).to(handle.device)
    return mean, sd, md, sdd


def _sweep(handle, kind, md, sdd, p0, p1, shape):
    out = torch.empty_like(md)
    _lib.check(handle.lib.gpimhip_acq(handle.h, _lib.ACQ_IDS[kind], _lib.ptr(md), _lib.ptr(sdd), md.numel(),
                                      float(p0), float(p1), None, _lib.ptr(out)))
    return out.cpu().numpy().reshape(shape), out


def _nanmax(handle, *tensors):
    t = tensors[0] if len(tensors) == 1 else torch.cat([x.reshape(-1) for x in tensors])
    t = t.contiguous()
    out = torch.empty((1,), dtype=_F64, device=t.device)
    _lib.check(handle.lib.gpimhip_nanmax(handle.h, _lib.ptr(t), t.numel(), _lib.ptr(out)))
    return out.item()


# ------------------------------------------------------------------ device-resident path (boptimizer)
def acquisition_on_device(gpmodel, kind, X_full, X_sparse, p0=0.0, p1=1.0, xi=0.01, Xf_d=None, mask_d=None,
                          Xobs_d=None):
    """The three built-in acquisition functions for a ``gpim_amd.reconstructor`` surrogate WITHOUT leaving
    the GPU: returns device tensors (acq, mean, sd) over the flattened grid (acq already multiplied by
    ``mask_d`` when one is given).  Same arithmetic as the public functions below.  The incumbent of EI / POI --
    nanmax of the posterior at the observed points, which the reference obtains from a second predict over the
    whole NaN-masked grid (acqfunc.py:58-59,86-88) -- is predicted at the observed rows only: every column of a
    prediction is independent of the others, so the values (and their nanmax) are the same and the N^2 M
    triangular product shrinks to N^2 N_obs.  Exact models go through ONE library call (gpimhip_acquire_exact:
    one factorisation, incumbent kept on the device, fused posterior + acquisition launch for N <= 384).
    ``Xobs_d``: the observed rows of ``X_sparse`` when the caller already holds them on the device (boptimizer:
    they are the surrogate's training inputs)."""
    handle = gpmodel._handle
    Xf = Xf_d if Xf_d is not None else gpmodel._to_device(_rows(X_full, gpmodel.precision))
    Xobs = Xobs_d
    if kind != "cb" and Xobs is None:
        Xs = _rows(X_sparse, gpmodel.precision)
        obs = ~torch.isnan(Xs).any(dim=1)
        Xobs = gpmodel._to_device(Xs[obs].contiguous())
    if not gpmodel.do_sparse and not gpmodel.do_structured:
        gpmodel._check_data()
        M = Xf.shape[0]
        mean_d = torch.empty((M,), dtype=_F64, device=Xf.device)
        sd_d = torch.empty_like(mean_d)
        out = torch.empty_like(mean_d)
        a, b = (p0, p1) if kind == "cb" else (0.0, xi)
        _lib.check(handle.lib.gpimhip_acquire_exact(
            handle.h, ctypes.byref(gpmodel._mstruct), _lib.ptr(gpmodel._Xd), _lib.ptr(gpmodel._yd), gpmodel._Xd.shape[0],
            _lib.ptr(gpmodel._u), _lib.ptr(Xf), M, None if Xobs is None else _lib.ptr(Xobs),
            0 if Xobs is None else Xobs.shape[0], _lib.ACQ_IDS[kind], float(a), float(b),
            None if mask_d is None else _lib.ptr(mask_d), _lib.ptr(mean_d), _lib.ptr(sd_d), _lib.ptr(out)))
        return out, mean_d, sd_d
    mean_d, sd_d = gpmodel._predict_device(Xf)
    if kind == "cb":
        a, b = p0, p1
    else:
        mo, so = gpmodel._predict_device(Xobs)
        a = _nanmax(handle, mo) if kind == "ei" else _nanmax(handle, mo, so)
        b = xi
    out = torch.empty_like(mean_d)
    _lib.check(handle.lib.gpimhip_acq(handle.h, _lib.ACQ_IDS[kind], _lib.ptr(mean_d), _lib.ptr(sd_d), mean_d.numel(),
                                      float(a), float(b), None if mask_d is None else _lib.ptr(mask_d), _lib.ptr(out)))
    return out, mean_d, sd_d


def _rows(X, precision):
    from . import gprutils
    return gprutils.prepare_test_data(np.asarray(X), precision=precision)


def confidence_bound(gpmodel, X_full, **kwargs):
    """alpha * mean + beta * sd (defaults 0, 1)."""
    alpha, beta = kwargs.get("alpha", 0), kwargs.get("beta", 1)
    handle = _handle_for(gpmodel)
    mean, sd, md, sdd = _device_pred(gpmodel, X_full, handle)
    acq, acq_d = _sweep(handle, "cb", md, sdd, alpha, beta, mean.shape)
    gpmodel._last_acq = acq_d if hasattr(gpmodel, "_last_pred") else None
    return acq, (mean, sd)


def expected_improvement(gpmodel, X_full, X_sparse, **kwargs):
    """imp * Phi(imp/sd) + sd * phi(imp/sd), imp = mean - max(mean at observed points) - xi."""
    xi = kwargs.get("xi", 0.01)
    handle = _handle_for(gpmodel)
    mean, sd, md, sdd = _device_pred(gpmodel, X_full, handle)
    _, _, mo, _ = _device_pred(gpmodel, X_sparse, handle)     # NaN rows -> NaN means
    best = _nanmax(handle, mo)
    acq, acq_d = _sweep(handle, "ei", md, sdd, best, xi, mean.shape)
    gpmodel._last_acq = acq_d if hasattr(gpmodel, "_last_pred") else None
    return acq, (mean, sd)


def probability_of_improvement(gpmodel, X_full, X_sparse, **kwargs):
    """Phi((mean - incumbent - xi) / sd) with the reference's (mean, sd)-tuple incumbent."""
    xi = kwargs.get("xi", 0.01)
    handle = _handle_for(gpmodel)
    mean, sd, md, sdd = _device_pred(gpmodel, X_full, handle)
    _, _, mo, so = _device_pred(gpmodel, X_sparse, handle)
    best = _nanmax(handle, mo, so)
    acq, acq_d = _sweep(handle, "poi", md, sdd, best, xi, mean.shape)
    gpmodel._last_acq = acq_d if hasattr(gpmodel, "_last_pred") else None
    return acq, (mean, sd)
