"""
Structured (Kronecker) exact GP on fully observed regular grids -- csrc/kron.hip, SURVEY 8(f) rank 3 --
against the DENSE oracle (O.ExactGP / O.reconstructor on the same inputs): the structured solver is the
same model solved with different algebra, so loss, gradient, Adam trajectory and posterior must agree
with the dense restatement up to rounding.  Also: the structured and the dense HIP paths against each
other at a size the oracle would take minutes for, and the ``skreconstructor`` class surface.
"""
import ctypes

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

from oracle import gpim_oracle as O
from problems import ckpfm_cube, oracle_threads


@pytest.fixture(scope="module")
def gpim(ensure_built):
    import gpim_amd
    return gpim_amd


def smooth_grid(shape, seed):
    rng = np.random.default_rng(seed)
    idx = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
    f = np.ones(shape)
    for k, g in enumerate(idx):
        f = f * np.cos(g / (2.0 + k) + 0.3 * k)
    return f + 0.05 * rng.standard_normal(shape)


@pytest.mark.parametrize("shape,iso", [((12, 10), False), ((9, 14), True), ((6, 5, 8), False), ((4, 3, 5, 4), False),
                                       ((1, 7), False), ((33, 20), False)])
def test_kron_loss_grad_predict_vs_dense_oracle(gpim, shape, iso):
    from gpim_amd import _lib
    from gpim_amd.kernels import KernelSpec
    d = len(shape)
    R = smooth_grid(shape, seed=sum(shape))
    Xg = gpim.utils.get_full_grid(R)
    X, y = gpim.utils.prepare_training_data(Xg, R)
    ls = [0.7, 9.0] if iso else [[0.7] * d, [9.0] * d]
    torch.manual_seed(2)
    kp = O.KernelParams("RBF", d, ls)
    spec = KernelSpec("RBF", d, ls, jitter=1e-5)
    u = spec.draw_initial_u(torch.Generator().manual_seed(2))
    with torch.no_grad():
        kp.u_noise.fill_(-2.5)
    u[1 + spec.n_ls] = -2.5
    gp = O.ExactGP(X, y, kp, 1e-5)
    loss_ref, g_ref = gp.loss_and_grad()
    H = _lib.Handle()
    m = spec.struct()
    axes = [np.arange(n, dtype=np.float64) for n in shape]
    n_arr = (ctypes.c_int32 * d)(*shape)
    axes_d = torch.from_numpy(np.concatenate(axes)).cuda()
    yd, ud = y.cuda().contiguous(), u.cuda()
    out = torch.empty(1 + spec.n_params, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_kron_nll_grad(H.h, ctypes.byref(m), d, n_arr, _lib.ptr(axes_d), _lib.ptr(yd), _lib.ptr(ud),
                                           ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    o = out.cpu()
    assert_allclose(o[0].item(), loss_ref.item(), rtol=1e-11)
    assert_allclose(o[1:].numpy(), g_ref.numpy(), rtol=1e-8, atol=1e-9 * max(1.0, g_ref.abs().max().item()))
    # prediction on a finer product grid that is not the training grid
    taxes = [np.linspace(-0.5, n - 0.5, 2 * n + 1) for n in shape]
    tn = (ctypes.c_int32 * d)(*[len(t) for t in taxes])
    M = int(np.prod([len(t) for t in taxes]))
    mean = torch.empty(M, dtype=torch.float64, device="cuda")
    var = torch.empty_like(mean)
    taxes_d = torch.from_numpy(np.concatenate(taxes)).cuda()
    _lib.check(H.lib.gpimhip_predict_kron(H.h, ctypes.byref(m), d, n_arr, _lib.ptr(axes_d), _lib.ptr(yd), _lib.ptr(ud),
                                          tn, _lib.ptr(taxes_d), _lib.ptr(mean), _lib.ptr(var)))
    Xs = torch.from_numpy(np.stack(np.meshgrid(*taxes, indexing="ij"), -1).reshape(-1, d))
    mref, vref = gp.predict(Xs)
    assert_allclose(mean.cpu().numpy(), mref.numpy(), rtol=0, atol=1e-9)
    assert_allclose(var.cpu().numpy(), vref.numpy(), rtol=0, atol=1e-9)
    H.close()


@pytest.mark.parametrize("shape", [(16, 12), (6, 5, 8)])
def test_structured_reconstructor_run_vs_dense_oracle(gpim, shape):
    """reconstructor(structured=True).run(): 60 Adam iterations + posterior follow the dense oracle."""
    R = smooth_grid(shape, seed=7)
    Xf = gpim.utils.get_full_grid(R)
    kw = dict(kernel="RBF", learning_rate=0.05, iterations=60, verbose=0)
    rec = gpim.reconstructor(Xf, R, Xf, structured=True, **kw)
    mean, sd, hyper = rec.run()
    mo, so, ho = O.reconstructor(Xf, R, Xf, **kw).run()
    assert mean.shape == R.shape
    assert_allclose(hyper["variance"], ho["variance"], rtol=1e-7)
    assert_allclose(hyper["lengthscale"], ho["lengthscale"], rtol=1e-7)
    assert_allclose(hyper["noise"], ho["noise"], rtol=1e-7)
    assert_allclose(mean, mo, atol=1e-8)
    assert_allclose(sd, so, atol=1e-8)
    # second train() call warm-starts; predict on a denser grid
    rec.train(iterations=5)
    assert len(rec.hyperparams["noise"]) == 65
    m2, s2 = rec.predict(gpim.utils.get_full_grid(R, dense_x=0.5))
    assert m2.shape == tuple(2 * n for n in shape) and np.isfinite(m2).all() and np.isfinite(s2).all()


def test_structured_equals_dense_hip_on_c5_slice(gpim):
    """One per-Ns slice of the C5 twin (10 x 10 x 64, N = 6400, fully observed): the structured solver
    against the dense HIP path (itself oracle-checked at this size in tests/test_gpu_regimes.py) --
    20 Adam iterations and the posterior on the grid."""
    R = ckpfm_cube()[..., 0]
    Xf = gpim.utils.get_full_grid(R)
    kw = dict(kernel="RBF", learning_rate=0.05, iterations=20, verbose=0)
    ms, ss, hs = gpim.reconstructor(Xf, R, Xf, structured=True, **kw).run()
    md, sdd, hd = gpim.reconstructor(Xf, R, Xf, **kw).run()
    assert_allclose(hs["lengthscale"], hd["lengthscale"], rtol=1e-7)
    assert_allclose(hs["noise"], hd["noise"], rtol=1e-7)
    assert_allclose(hs["variance"], hd["variance"], rtol=1e-7)
    assert_allclose(ms, md, atol=1e-7)
    assert_allclose(ss, sdd, atol=1e-7)


def test_full_image_256_structured(gpim):
    """A complete 256 x 256 image (N = 65536): out of reach for the dense path's 32 GiB covariance on a
    test budget, a few milliseconds per iteration here.  Properties: loss decreases, the posterior
    reproduces the image within the noise level, sd is at the noise floor everywhere (all pixels
    observed); agreement with the dense oracle is established at small sizes above."""
    R = smooth_grid((256, 256), seed=1)
    Xf = gpim.utils.get_full_grid(R)
    rec = gpim.skreconstructor(Xf, R, Xf, kernel="RBF", lengthscale=[[1., 1.], [40., 40.]], learning_rate=0.1,
                               iterations=30, verbose=0)
    mean, sd, hyper = rec.run()
    assert mean.shape == R.shape and np.isfinite(mean).all() and np.isfinite(sd).all()
    loss = np.array(rec.loss_all)
    assert loss[-1] < loss[0]
    assert np.sqrt(np.mean((mean - R) ** 2)) < 3 * np.sqrt(hyper["noise"][-1]) + 0.05
    assert set(["lengthscale", "noise"]) <= set(hyper)


def test_structured_rejects_what_it_cannot_do(gpim):
    R = smooth_grid((8, 8), seed=3)
    Xf = gpim.utils.get_full_grid(R)
    # (Matern52 / RationalQuadratic on a complete grid go through the reflection blocks since round 5: test_gpu_symm.py)
    Rn = R.copy()
    Rn[2, 3] = np.nan
    with pytest.raises(NotImplementedError):
        gpim.reconstructor(gpim.utils.get_sparse_grid(Rn), Rn, Xf, structured=True, verbose=0)
    Xbad = Xf.copy()
    Xbad[0, 3, 4] += 0.5
    with pytest.raises(NotImplementedError):
        gpim.reconstructor(Xbad, R, Xf, structured=True, verbose=0)
