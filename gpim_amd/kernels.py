"""
kernels.py -- kernel family + hyper-parameter parameterisation (host side).

Mirror of the reference's gpim/kernels/pyro_kernels.py:14-96 ``get_kernel`` (SURVEY 8(a)
row a4): an RBF / Matern52 / RationalQuadratic kernel whose variance and lengthscale carry
Uniform priors, which Pyro turns into interval-constrained MAP parameters INITIALISED BY ONE
DRAW FROM EACH PRIOR (variance first, then lengthscale) with torch's CPU generator.  The
draw is reproduced here bit for bit; the constrained<->unconstrained maps and all kernel
arithmetic live on the device (csrc/engine.hip, csrc/kfun.hpp).
"""
import ctypes
import math

import torch

from . import _lib

_F64 = torch.float64
_TINY = torch.finfo(_F64).tiny
_EPS = torch.finfo(_F64).eps


def _logit_clipped(p):
    p = p.clamp(min=_TINY, max=1.0 - _EPS)
    return p.log() - (-p).log1p()


class KernelSpec:
    """What ``get_kernel`` returns: bounds, the initial unconstrained vector u0 and the
    gpimhip_model_t that the device code consumes."""

    def __init__(self, kernel_type, input_dim, lengthscale, amplitude=None, jitter=1e-5):
        if kernel_type not in _lib.KERNEL_IDS:
            print('Select one of the currently available kernels:',
                  '"RBF", "RationalQuadratic", "Matern52"')
            raise KeyError(kernel_type)
        if not 1 <= input_dim <= _lib.MAX_DIM:
            raise NotImplementedError("input dimensionality must be between 1 and 4")
        self.kernel_type = kernel_type
        self.dim = int(input_dim)
        amp = [1e-4, 10.] if amplitude is None else amplitude
        self.amp_lo, self.amp_hi = float(amp[0]), float(amp[1])
        lo = torch.as_tensor(lengthscale[0], dtype=_F64)
        hi = torch.as_tensor(lengthscale[1], dtype=_F64)
        self.isotropic = lo.dim() == 0
        self.ls_lo = lo.reshape(-1).clone()
        self.ls_hi = hi.reshape(-1).clone()
        self.n_ls = int(self.ls_lo.numel())
        if self.n_ls not in (1, self.dim):
            raise ValueError("lengthscale bounds must be two scalars or two lists of length input_dim")
        self.jitter = float(jitter)
        self.n_params = 2 + self.n_ls + (1 if kernel_type == "RationalQuadratic" else 0)

    def draw_initial_u(self, generator=None, dtype=_F64):
        """The two prior draws mapped to u.  generator: a torch CPU Generator (default: the global
        one, which the caller has just seeded like gpr.py:101 does); a private generator seeded
        with the same value yields the same numbers and is safe under threads.  dtype: the
        precision the reference would draw in (float32 for precision='single'); u itself is
        always float64."""
        alo, ahi = torch.tensor(self.amp_lo, dtype=dtype), torch.tensor(self.amp_hi, dtype=dtype)
        v0 = (alo + torch.rand((), dtype=dtype, generator=generator) * (ahi - alo)).to(_F64)
        shape = () if self.isotropic else (self.n_ls,)
        lo_, hi_ = self.ls_lo.reshape(shape).to(dtype), self.ls_hi.reshape(shape).to(dtype)
        l0 = (lo_ + torch.rand(shape, dtype=dtype, generator=generator) * (hi_ - lo_)).to(_F64)
        alo, ahi = alo.to(_F64), ahi.to(_F64)
        u = torch.zeros(self.n_params, dtype=_F64)          # noise = exp(0) = 1, alpha_rq = exp(0) = 1
        u[0] = _logit_clipped((v0 - alo) / (ahi - alo))
        u[1:1 + self.n_ls] = _logit_clipped((l0.reshape(-1) - self.ls_lo) / (self.ls_hi - self.ls_lo))
        return u

    def struct(self):
        m = _lib.ModelStruct()
        m.kernel = _lib.KERNEL_IDS[self.kernel_type]
        m.dim, m.n_ls, m.reserved = self.dim, self.n_ls, 0
        m.amp_lo, m.amp_hi = self.amp_lo, self.amp_hi
        for k in range(_lib.MAX_DIM):
            m.ls_lo[k] = float(self.ls_lo[k]) if k < self.n_ls else 0.0
            m.ls_hi[k] = float(self.ls_hi[k]) if k < self.n_ls else 1.0
        m.jitter = self.jitter
        return m

    # constrained values from a host copy of u (reporting only; the device has its own map)
    def constrained(self, u):
        u = u.detach().to("cpu", _F64)
        s = torch.clamp(torch.sigmoid(u[:1 + self.n_ls]), min=_TINY, max=1.0 - _EPS)
        var = self.amp_lo + (self.amp_hi - self.amp_lo) * s[0]
        ls = self.ls_lo + (self.ls_hi - self.ls_lo) * s[1:]
        noise = torch.exp(u[1 + self.n_ls])
        return var, ls, noise


def get_kernel(kernel_type, input_dim, lengthscale, use_gpu=False, **kwargs):
    """Same call shape as the reference's pyro_kernels.get_kernel; ``use_gpu`` is accepted
    and ignored (the engine is GPU-only); ``precision`` only affects the dtype of the initial
    draw (see reconstructor)."""
    return KernelSpec(kernel_type, input_dim, lengthscale, amplitude=kwargs.get("amplitude"),
                      jitter=kwargs.get("jitter", 1e-5))
