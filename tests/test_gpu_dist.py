"""
One exact GP across GPUs -- gpim_amd/dist_chol.py on the HIP tile engine, single process (P = 1): the
block-column-cyclic driver (right-looking by 512-column panels) against gpimhip_potrf (left-looking step schedule
below N = 12288: a different summation order of the same tile products) and torch.linalg.cholesky, entry-wise to
1e-12 of the largest entry; log det / solve / NLL / posterior mean against torch and the dense oracle.  The multi-rank
schedule itself (ownership, broadcasts, small collectives of the solves) is covered on CPU ranks in
tests/test_dist_gloo.py with a stub engine; a multi-GPU run needs the driver's 8-GPU node.
"""
import ctypes

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

from oracle import gpim_oracle as O


@pytest.mark.parametrize("n", [700, 1500, 4096])
def test_p1_factor_vs_potrf(ensure_built, n):
    from gpim_amd import _lib
    from gpim_amd.dist_chol import DistributedCholesky
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n // 3))
    A = torch.from_numpy(B @ B.T + n * np.eye(n)).cuda()
    ch = DistributedCholesky(n)
    ch.set_from_function(lambda c0, c1: A[:, c0:c1]).factor()
    Ld = ch.gather_lower()
    H = _lib.Handle()
    Lp = A.clone()
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(H.lib.gpimhip_potrf(H.h, _lib.ptr(Lp), n, n, _lib.ptr(info)))
    torch.cuda.synchronize()
    assert info.item() == 0
    ref = torch.linalg.cholesky(A)
    scale = ref.abs().max().item()
    assert (Ld - torch.tril(Lp)).abs().max().item() <= 1e-12 * scale
    assert (Ld - ref).abs().max().item() <= 1e-12 * scale
    assert_allclose(ch.logdet(), 2 * torch.log(torch.diagonal(ref)).sum().item(), rtol=1e-12)
    y = torch.from_numpy(rng.standard_normal(n)).cuda()
    alpha = ch.solve(y)
    assert_allclose(alpha.cpu().numpy(), torch.cholesky_solve(y[:, None], ref)[:, 0].cpu().numpy(), rtol=1e-8, atol=1e-12)
    H.close()


def test_grouped_kinv_pass_equals_one_panel_per_launch(ensure_built, monkeypatch):
    """K^-1 = X^T X with eight panels per launch (gpimhip_dist_kinv_update_n, DistributedCholesky._stream_groups) against one
    panel per launch: the same tiles with the same k-ranges, hence the same bits; against torch to 1e-10.  n = 5300: eleven
    panels, a group of eight and a ragged group of three."""
    from gpim_amd import dist_chol
    from gpim_amd.dist_chol import DistributedCholesky
    n = 5300
    rng = np.random.default_rng(n)
    Bm = rng.standard_normal((n, n // 3))
    Ah = torch.from_numpy(Bm @ Bm.T + n * np.eye(n))
    A = Ah.cuda()
    outs = []
    for G in (1, 8, 3):
        monkeypatch.setattr(dist_chol, "KINV_GROUP", G)
        ch = DistributedCholesky(n)
        ch.set_from_function(lambda c0, c1: A[:, c0:c1]).factor()
        outs.append(torch.tril(ch.kinv(ch.inverse())[:n, :n]).clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    ref = torch.linalg.inv(Ah)
    assert (outs[1].cpu() - torch.tril(ref)).abs().max().item() <= 1e-10 * ref.abs().max().item()


def test_not_pd_raises(ensure_built):
    from gpim_amd.dist_chol import DistributedCholesky
    A = torch.eye(900, dtype=torch.float64, device="cuda")
    A[650, 650] = -2.0
    with pytest.raises(torch.linalg.LinAlgError):
        DistributedCholesky(900).set_from_function(lambda c0, c1: A[:, c0:c1]).factor()


def test_exact_gp_posterior_mean_vs_oracle(ensure_built):
    """Posterior mean, sd and NLL of one exact GP through the distributed driver (P = 1) against the dense oracle."""
    from gpim_amd.dist_chol import exact_gp_posterior
    rng = np.random.default_rng(2)
    pts = np.unique(rng.integers(0, 60, size=(4000, 2)), axis=0)
    pts = pts[rng.permutation(len(pts))[:1300]].astype(np.float64)
    y = np.sin(pts[:, 0] / 7.0) * np.cos(pts[:, 1] / 9.0) + 0.05 * rng.standard_normal(len(pts))
    Xs = rng.uniform(0, 59, size=(777, 2))
    ls, var, noise = [6.0, 8.0], 1.3, 0.02
    mean, sd, nll = exact_gp_posterior(pts, y, Xs, kernel="Matern52", lengthscale=ls, variance=var, noise=noise,
                                       chunk_bytes=300 * 8 * 1408)          # three chunks of test columns
    kp = O.KernelParams("Matern52", 2, [[0., 0.], [12., 16.]])
    with torch.no_grad():
        kp.u_var.copy_(torch.logit(torch.tensor((var - 1e-4) / (10 - 1e-4), dtype=torch.float64)))
        kp.u_ls.copy_(torch.zeros(2, dtype=torch.float64))          # sigmoid(0) = 1/2 -> ls = hi / 2
        kp.u_noise.copy_(torch.log(torch.tensor(noise, dtype=torch.float64)))
    gp = O.ExactGP(torch.from_numpy(pts), torch.from_numpy(y), kp, 1e-5)
    mref, vref = gp.predict(torch.from_numpy(Xs))
    assert_allclose(mean, mref.numpy(), atol=1e-9)
    assert_allclose(sd, vref.sqrt().numpy(), atol=1e-9)
    assert_allclose(nll, (gp.loss() - kp.neg_log_prior()).item(), rtol=1e-11)


def _train_problem(n=1300, seed=5):
    rng = np.random.default_rng(seed)
    pts = np.unique(rng.integers(0, 64, size=(6000, 2)), axis=0)
    pts = pts[rng.permutation(len(pts))[:n]].astype(np.float64)
    y = np.sin(pts[:, 0] / 6.0) * np.cos(pts[:, 1] / 8.0) + 0.05 * rng.standard_normal(len(pts))
    return pts, y


@pytest.mark.parametrize("kind", ["RBF", "Matern52"])
def test_distributed_loss_and_gradient_vs_single_gpu(ensure_built, kind):
    """Loss and d loss / du through the distributed passes (K columns, block-column-cyclic factor, streamed inverse,
    K^-1 = X^T X, sharded gradient sums) at P = 1 against gpimhip_nll_grad (pinned to the oracle in test_gpu_ops.py)."""
    from gpim_amd import _lib
    from gpim_amd.dist_chol import exact_gp_nll_grad
    from gpim_amd.kernels import KernelSpec
    pts, y = _train_problem()
    ls = [[1., 1.], [20., 20.]]
    spec = KernelSpec(kind, 2, ls, jitter=1e-5)
    u = spec.draw_initial_u(torch.Generator().manual_seed(3))
    loss_d, grad_d = exact_gp_nll_grad(pts, y, u, kernel=kind, lengthscale=ls)
    H = _lib.Handle()
    m = spec.struct()
    Xd, yd, ud = (torch.from_numpy(a).cuda() for a in (pts, y, u.numpy()))
    out = torch.zeros(1 + spec.n_params, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(y), _lib.ptr(ud),
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    o = out.cpu().numpy()
    assert_allclose(loss_d, o[0], rtol=1e-11)
    assert_allclose(grad_d, o[1:], rtol=1e-8, atol=1e-9 * np.abs(o[1:]).max())
    H.close()


def test_distributed_training_vs_reconstructor(ensure_built):
    """exact_gp_fit (P = 1) follows reconstructor.train: same seed -> same initial draw, same Adam trajectory."""
    import gpim_amd
    from gpim_amd.dist_chol import exact_gp_fit
    rng = np.random.default_rng(11)
    R = np.full((48, 48), np.nan)
    idx = rng.permutation(48 * 48)[:900]
    ii, jj = np.unravel_index(idx, R.shape)
    R[ii, jj] = np.sin(ii / 5.0) * np.cos(jj / 7.0) + 0.05 * rng.standard_normal(len(idx))
    X, Xf = gpim_amd.utils.get_sparse_grid(R), gpim_amd.utils.get_full_grid(R)
    kw = dict(kernel="Matern52", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=8)
    rec = gpim_amd.reconstructor(X, R, Xf, verbose=0, seed=0, **kw)
    rec.train()
    pts = rec.X.cpu().numpy()
    yv = rec.y.cpu().numpy()
    hyper, u = exact_gp_fit(pts, yv, seed=0, **kw)
    assert_allclose(hyper["lengthscale"], np.asarray(rec.hyperparams["lengthscale"]), rtol=1e-8)
    assert_allclose(hyper["variance"], np.asarray(rec.hyperparams["variance"]), rtol=1e-8)
    assert_allclose(hyper["noise"], np.asarray(rec.hyperparams["noise"]), rtol=1e-8)
