"""
Sparse (inducing-point, VFE) GP on the MI355X -- SURVEY 8(a) row a16 -- against the oracle's
restatement of pyro.contrib.gp.models.SparseGPRegression (torch CPU fp64 + autograd).  The reference
holds no known answer for this model (parity unpinned; only shape/NaN smoke in its own tests), so
the oracle is the only checker: operator level (loss, full gradient incl. the inducing inputs,
posterior) and end to end through ``reconstructor(sparse=True)``.
"""
import ctypes

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

from oracle import gpim_oracle as O
from problems import spiral_image


@pytest.fixture(scope="module")
def gpim(ensure_built):
    import gpim_amd
    return gpim_amd


@pytest.mark.parametrize("kind,N,Mu,d", [("RBF", 200, 20, 2), ("Matern52", 700, 150, 2), ("RBF", 1000, 260, 3),
                                         ("RationalQuadratic", 300, 40, 2)])
def test_vfe_loss_grad_predict(gpim, kind, N, Mu, d):
    from gpim_amd import _lib
    from gpim_amd.kernels import KernelSpec
    H = _lib.Handle()
    rng = np.random.default_rng(N)
    X = torch.from_numpy(np.unique(rng.integers(0, 40, size=(4 * N, d)), axis=0)[:N].astype(np.float64))
    N = len(X)
    y = torch.from_numpy(np.sin(X.numpy().sum(1) / 6.0) + 0.1 * rng.standard_normal(N))
    ls = [[1.0] * d, [15.0] * d]
    torch.manual_seed(1)
    kp = O.KernelParams(kind, d, ls)
    spec = KernelSpec(kind, d, ls, jitter=1e-5)
    u_t = spec.draw_initial_u(torch.Generator().manual_seed(1))
    with torch.no_grad():
        kp.u_noise.fill_(-2.0)
    u_t[1 + spec.n_ls] = -2.0
    Xu0 = X[::N // Mu].clone()
    Mu = len(Xu0)
    gp = O.SparseGP(X, y, kp, Xu0, 1e-5)
    loss_ref, g_ref = gp.loss_and_grad()
    u = torch.cat([u_t, Xu0.reshape(-1)]).cuda()
    m = spec.struct()
    Xd, yd = X.cuda().contiguous(), y.cuda().contiguous()
    P = spec.n_params
    out = torch.empty(1 + P + Mu * d, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_vfe_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(u),
                                          ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    o = out.cpu()
    assert_allclose(o[0].item(), loss_ref.item(), rtol=1e-12)
    assert_allclose(o[1:1 + P].numpy(), g_ref[:P].numpy(), rtol=1e-9, atol=1e-9)
    assert_allclose(o[1 + P:].numpy(), g_ref[P:].numpy(), rtol=0, atol=1e-9 * max(1.0, g_ref[P:].abs().max().item()))
    Xs = torch.from_numpy(rng.uniform(0, 40, size=(500, d)))
    Xsd = Xs.cuda().contiguous()
    mean = torch.empty(500, dtype=torch.float64, device="cuda")
    var = torch.empty_like(mean)
    _lib.check(H.lib.gpimhip_predict_vfe(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(u),
                                         _lib.ptr(Xsd), 500, _lib.ptr(mean), _lib.ptr(var)))
    mr, vr = gp.predict(Xs)
    assert_allclose(mean.cpu().numpy(), mr.numpy(), atol=1e-10)
    assert_allclose(var.cpu().numpy(), vr.numpy(), atol=1e-10)
    H.close()


@pytest.mark.parametrize("kernel", ["RBF", "Matern52"])
def test_sparse_reconstructor_run(gpim, kernel):
    """reconstructor(sparse=True).run(): hyper-parameter and inducing-point histories and the
    reconstruction follow the oracle over 40 Adam iterations."""
    R, _ = spiral_image(size=48, keep=0.3, seed=2)
    X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
    kw = dict(kernel=kernel, lengthscale=[[1., 1.], [8., 8.]], sparse=True, indpoints=60, learning_rate=0.05,
              iterations=40, verbose=0)
    rec = gpim.reconstructor(X, R, Xf, **kw)
    mean, sd, hyper = rec.run()
    orc = O.reconstructor(X, R, Xf, **kw)
    mo, so, ho = orc.run()
    assert mean.shape == R.shape and not np.isnan(mean).any() and not np.isnan(sd).any()
    assert len(hyper["inducing_points"]) == 40 and hyper["inducing_points"][0].shape == ho["inducing_points"][0].shape
    assert_allclose(hyper["variance"], ho["variance"], rtol=1e-7)
    assert_allclose(hyper["lengthscale"], ho["lengthscale"], rtol=1e-7)
    assert_allclose(hyper["noise"], ho["noise"], rtol=1e-7)
    assert_allclose(hyper["inducing_points"][-1], ho["inducing_points"][-1], atol=1e-7)
    assert_allclose(mean, mo, atol=1e-7)
    assert_allclose(sd, so, atol=1e-7)
    assert rec.model.Xu.shape == (rec._n_ind, 2)


def test_sparse_default_indpoints_and_3d(gpim):
    """indpoints=None -> N // 10 (gpr.py:146-148); 3D input; second train() call warm-starts."""
    rng = np.random.default_rng(3)
    R3 = np.sin(np.arange(10)[:, None, None] / 3.0) * np.cos(np.arange(9)[None, :, None] / 2.0) * \
        np.ones((1, 1, 6)) + 0.01 * rng.standard_normal((10, 9, 6))
    R3[rng.random((10, 9)) < 0.4] = np.nan
    X, Xf = gpim.utils.get_sparse_grid(R3), gpim.utils.get_full_grid(R3)
    kw = dict(kernel="RBF", sparse=True, learning_rate=0.05, iterations=15, verbose=0)
    rec = gpim.reconstructor(X, R3, Xf, **kw)
    orc = O.reconstructor(X, R3, Xf, **kw)
    n = rec.X.shape[0]
    assert rec._n_ind == len(rec.X[::n // (n // 10)]) == orc.model.Xu.shape[0]
    rec.train()
    rec.train(iterations=5)
    orc.train()
    orc.train(iterations=5)
    assert len(rec.hyperparams["inducing_points"]) == 20
    assert_allclose(rec.hyperparams["noise"], orc.hyperparams["noise"], rtol=1e-7)
    m, s = rec.predict()
    mo, so = orc.predict()
    assert_allclose(m, mo, atol=1e-7)
    assert_allclose(s, so, atol=1e-7)


def test_c5_shaped_4d_cube_slices_sparse(gpim):
    """Config C5 in miniature: a 4D cube, one sparse-VFE GP per slice along the last axis (3D slices,
    fully observed), each equal to a stand-alone reconstructor(sparse=True) and close to the oracle."""
    from gpim_amd import dist as gd
    rng = np.random.default_rng(5)
    i, j, v = np.meshgrid(np.arange(6), np.arange(5), np.arange(8), indexing="ij")
    cube = np.stack([np.sin(i / 2.0 + s) * np.cos(j / 2.0) * np.exp(-((v - 4.0) / 3.0) ** 2) for s in range(3)], -1)
    cube = cube + 0.01 * rng.standard_normal(cube.shape)
    kw = dict(kernel="RBF", sparse=True, indpoints=24, learning_rate=0.05, iterations=20)
    mean, sd = gd.reconstruct_slices(cube, axis=-1, **kw)
    assert mean.shape == cube.shape and np.isfinite(mean).all() and np.isfinite(sd).all()
    R = cube[..., 1]
    Xf = gpim.utils.get_full_grid(R)
    m1, s1, _ = gpim.reconstructor(Xf, R, Xf, verbose=0, **kw).run()
    np.testing.assert_array_equal(mean[..., 1], m1)      # (the slices run as a lock-step batch: the same bits)
    mo, so, _ = O.reconstructor(Xf, R, Xf, verbose=0, **kw).run()
    assert_allclose(mean[..., 1], mo, atol=1e-7)
    assert_allclose(sd[..., 1], so, atol=1e-7)


def test_sparse_lock_step_batch_equals_stand_alone_models(gpim):
    """gpimhip_fit_vfe_batched / gpimhip_predict_vfe_batched (B sparse models in every launch) against B stand-alone
    reconstructor(sparse=True) runs: histories of the hyper-parameters and of the inducing inputs, posterior -- the same
    bits (the tile engine picks its launch shapes from the tiles of one model, GemmArgs::shape_div)."""
    from gpim_amd.batch import fit_predict_batch_sparse
    rng = np.random.default_rng(11)
    i, j, v = np.meshgrid(np.arange(20), np.arange(18), np.arange(6), indexing="ij")
    Rs = [np.sin(i / 4.0 + 0.7 * b) * np.cos(j / 3.0) * np.exp(-((v - 2.5) / 2.0) ** 2) + 0.02 * rng.standard_normal(i.shape)
          for b in range(3)]
    Xf = gpim.utils.get_full_grid(Rs[0])
    kw = dict(kernel="Matern52", lengthscale=[[1., 1., 1.], [10., 10., 10.]], indpoints=300, learning_rate=0.05, iterations=12)
    mean, sd, hist, hist_xu = fit_predict_batch_sparse(Xf, Rs, Xf, **kw)
    assert hist.shape == (3, 12, 5) and hist_xu.shape[:2] == (3, 12)
    for b in (0, 2):
        rec = gpim.reconstructor(Xf, Rs[b], Xf, sparse=True, verbose=0, **kw)
        m1, s1, h1 = rec.run()
        assert hist_xu.shape[2] == rec._n_ind
        np.testing.assert_array_equal(hist[b, :, 0].cpu().numpy(), np.asarray(h1["variance"]))
        np.testing.assert_array_equal(hist[b, :, 1:4].cpu().numpy(), np.asarray(h1["lengthscale"]))
        np.testing.assert_array_equal(hist[b, :, 4].cpu().numpy(), np.asarray(h1["noise"]))
        np.testing.assert_array_equal(hist_xu[b, -1].cpu().numpy(), h1["inducing_points"][-1])
        np.testing.assert_array_equal(mean[b].cpu().numpy(), m1)
        np.testing.assert_array_equal(sd[b].cpu().numpy(), s1)


def test_sparse_slices_batched_and_one_by_one_agree(gpim):
    """dist.reconstruct_slices(sparse=True): the lock-step batch (default) and one reconstructor per slice (sparse_batch=0)."""
    from gpim_amd import dist as gd
    rng = np.random.default_rng(6)
    i, j, v = np.meshgrid(np.arange(7), np.arange(6), np.arange(9), indexing="ij")
    cube = np.stack([np.cos(i / 2.0 - s) * np.sin(j / 2.0 + 0.3) * np.exp(-((v - 4.0) / 3.0) ** 2) for s in range(5)], -1)
    cube = cube + 0.01 * rng.standard_normal(cube.shape)
    kw = dict(kernel="RBF", sparse=True, indpoints=40, learning_rate=0.05, iterations=15)
    mb, sb, hb = gd.reconstruct_slices(cube, axis=-1, return_hyperparams=True, **kw)
    m1, s1, h1 = gd.reconstruct_slices(cube, axis=-1, return_hyperparams=True, sparse_batch=0, **kw)
    np.testing.assert_array_equal(mb, m1)
    np.testing.assert_array_equal(sb, s1)
    for k in range(5):
        np.testing.assert_array_equal(np.asarray(hb[k]["noise"]), np.asarray(h1[k]["noise"]))
        np.testing.assert_array_equal(np.asarray(hb[k]["lengthscale"]), np.asarray(h1[k]["lengthscale"]))
        np.testing.assert_array_equal(hb[k]["inducing_points"][-1], h1[k]["inducing_points"][-1])


def test_sparse_lock_step_batch_with_different_inputs_per_model(gpim):
    """Slices with different NaN patterns of the same size: every model of the batch has its own inputs (x_stride != 0)
    and its own initial inducing inputs."""
    from gpim_amd.batch import fit_predict_batch_sparse
    rng = np.random.default_rng(21)
    i, j = np.meshgrid(np.arange(30), np.arange(28), indexing="ij")
    Rs = []
    for b in range(3):
        R = np.sin(i / 5.0 + b) * np.cos(j / 4.0) + 0.02 * rng.standard_normal(i.shape)
        R.ravel()[rng.permutation(R.size)[:240]] = np.nan          # 600 of 840 pixels observed, a different set per slice
        Rs.append(R)
    Xs = [gpim.utils.get_sparse_grid(R) for R in Rs]
    Xf = gpim.utils.get_full_grid(Rs[0])
    kw = dict(kernel="RBF", lengthscale=[[1., 1.], [10., 10.]], indpoints=150, learning_rate=0.05, iterations=10)
    mean, sd, hist, hist_xu = fit_predict_batch_sparse(Xs, Rs, Xf, **kw)
    for b in range(3):
        m1, s1, h1 = gpim.reconstructor(Xs[b], Rs[b], Xf, sparse=True, verbose=0, **kw).run()
        np.testing.assert_array_equal(hist[b, :, 3].cpu().numpy(), np.asarray(h1["noise"]))
        np.testing.assert_array_equal(hist_xu[b, -1].cpu().numpy(), h1["inducing_points"][-1])
        np.testing.assert_array_equal(mean[b].cpu().numpy(), m1)
        np.testing.assert_array_equal(sd[b].cpu().numpy(), s1)


def test_sparse_batch_non_pd_freezes_the_batch(gpim):
    """A factorisation that fails in ONE model of a lock-step batch (two identical inducing inputs, no jitter: k(Xu, Xu) is
    exactly singular) makes the batched fit return GPIMHIP_E_NOT_PD with no iteration completed, like the stand-alone
    call; the other model's parameters are left as they were."""
    from gpim_amd import _lib
    from gpim_amd.kernels import KernelSpec
    H = _lib.Handle()
    rng = np.random.default_rng(31)
    N, Mu, d, B = 300, 40, 2, 2
    X = torch.from_numpy(rng.uniform(0, 20, size=(N, d)))
    y = torch.from_numpy(np.sin(X.numpy().sum(1) / 4.0) + 0.05 * rng.standard_normal((B, N)))
    spec = KernelSpec("RBF", d, [[1.0] * d, [10.0] * d], jitter=0.0)
    u0 = spec.draw_initial_u(torch.Generator().manual_seed(0))
    Xu = X[::N // Mu][:Mu].clone()
    Xu_bad = Xu.clone()
    Xu_bad[7] = Xu_bad[3]                                # model 1: a repeated inducing input
    u = torch.stack([torch.cat([u0, Xu.reshape(-1)]), torch.cat([u0, Xu_bad.reshape(-1)])]).cuda().contiguous()
    u_before = u.clone()
    m = spec.struct()
    Xd, yd = X.cuda().contiguous(), y.cuda().contiguous()
    rc = H.lib.gpimhip_fit_vfe_batched(H.h, ctypes.byref(m), _lib.ptr(Xd), 0, _lib.ptr(yd), N, Mu, B, _lib.ptr(u), 0.05, 20,
                                       None, None, None)
    assert rc == _lib.E_NOT_PD
    assert int(H.lib.gpimhip_fit_completed(H.h)) == 0
    assert torch.equal(u, u_before)
    # the good model alone trains
    u1 = u_before[0].clone().contiguous()
    _lib.check(H.lib.gpimhip_fit_vfe(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(u1), 0.05, 20, None, None,
                                     None))
    assert torch.isfinite(u1).all() and not torch.equal(u1, u_before[0])
    H.close()
