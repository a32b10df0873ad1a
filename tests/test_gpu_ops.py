"""
Operator-level parity on the MI355X: every C-ABI entry point against the CPU oracle
(oracle/gpim_oracle.py) or torch-CPU LAPACK on the same seeded inputs.  fp64 throughout;
tolerances are stated per test (they are a few hundred ulps of the quantities involved --
the arithmetic is the same algebra in a different summation order).
"""
import ctypes

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose
from scipy.stats import norm

pytestmark = pytest.mark.gpu

from oracle import gpim_oracle as O


@pytest.fixture(scope="module")
def eng(ensure_built):
    from gpim_amd import _lib
    H = _lib.Handle()
    yield _lib, H
    H.close()


def scattered(N, d, seed, grid=24):
    rng = np.random.default_rng(seed)
    X = np.unique(rng.integers(0, grid, size=(N * 4, d)), axis=0).astype(np.float64)
    rng.shuffle(X)
    X = X[:N]
    y = np.sin(X.sum(1) / 5.0) + 0.1 * rng.standard_normal(len(X))
    return torch.from_numpy(np.ascontiguousarray(X)), torch.from_numpy(y)


def pair(kind, d, ls, seed, jitter=1e-5, noise_u=-3.0):
    """(oracle KernelParams, gpim_amd KernelSpec, u) holding identical parameters."""
    from gpim_amd.kernels import KernelSpec
    torch.manual_seed(seed)
    kp = O.KernelParams(kind, d, ls)
    torch.manual_seed(seed)
    spec = KernelSpec(kind, d, ls, jitter=jitter)
    u = spec.draw_initial_u()
    with torch.no_grad():
        kp.u_noise.fill_(noise_u)
    u[1 + spec.n_ls] = noise_u
    return kp, spec, u


@pytest.mark.parametrize("n", [1, 5, 16, 100, 128, 129, 300, 1000])
def test_potrf(eng, n):
    _lib, H = eng
    g = torch.Generator().manual_seed(n)
    B = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = B @ B.T / n + 0.5 * torch.eye(n, dtype=torch.float64)
    A[0, n - 1] = 123.0 if n > 1 else A[0, 0]          # strict upper part must stay untouched
    Ad = A.cuda().contiguous()
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(H.lib.gpimhip_potrf(H.h, _lib.ptr(Ad), n, n, _lib.ptr(info)))
    torch.cuda.synchronize()
    L = torch.linalg.cholesky(torch.tril(A) + torch.tril(A, -1).T)
    out = Ad.cpu()
    assert info.item() == 0
    assert_allclose(torch.tril(out).numpy(), L.numpy(), rtol=0, atol=5e-14)
    if n > 1:
        assert out[0, n - 1].item() == 123.0


@pytest.mark.parametrize("n", [300, 700, 1207, 2100])
def test_fused_panel_solve_and_diagonal_update_same_bits(eng, n, monkeypatch):
    """F_j and D_j as ONE launch (panel_solve_diag_kernel: the diagonal tile's workgroups solve the strips they need
    themselves, from the copy the step launch left) against the two launches (GPIMHIP_NO_FUSED_FD): the same bits in the
    factor, and in a lock-step batch that runs as two halves (exact fits, six problems)."""
    _lib, H = eng
    g = torch.Generator().manual_seed(n)
    Bm = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = Bm @ Bm.T / n + 0.5 * torch.eye(n, dtype=torch.float64)
    outs = []
    for knob in (None, "1"):
        if knob:
            monkeypatch.setenv("GPIMHIP_NO_FUSED_FD", knob)
        else:
            monkeypatch.delenv("GPIMHIP_NO_FUSED_FD", raising=False)
        Ad = A.cuda().contiguous()
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(H.lib.gpimhip_potrf(H.h, _lib.ptr(Ad), n, n, _lib.ptr(info)))
        torch.cuda.synchronize()
        assert info.item() == 0
        outs.append(torch.tril(Ad.cpu()))
    assert torch.equal(outs[0], outs[1])
    if n == 700:
        from gpim_amd.batch import fit_predict_batch
        rng = np.random.default_rng(3)
        ii, jj = np.meshgrid(np.arange(24), np.arange(20), indexing="ij")
        Rs = []
        for b in range(6):
            R = np.sin(ii / 4.0 + b) * np.cos(jj / 3.0) + 0.05 * rng.standard_normal(ii.shape)
            R.ravel()[rng.permutation(R.size)[:180]] = np.nan          # 300 points each: three block columns
            Rs.append(R)
        import gpim_amd
        Xs = [gpim_amd.utils.get_sparse_grid(R) for R in Rs]
        Xf = gpim_amd.utils.get_full_grid(Rs[0])
        res = []
        for knob in (None, "1"):
            if knob:
                monkeypatch.setenv("GPIMHIP_NO_FUSED_FD", knob)
            else:
                monkeypatch.delenv("GPIMHIP_NO_FUSED_FD", raising=False)
            mean, sd, hist = fit_predict_batch(Xs, Rs, Xf, kernel="RBF", lengthscale=[[1., 1.], [10., 10.]], learning_rate=0.1,
                                               iterations=6)
            res.append((mean.cpu(), sd.cpu(), hist.cpu()))
        for x, y in zip(res[0], res[1]):
            assert torch.equal(x, y)


@pytest.mark.parametrize("n,col", [(200, 150), (1, 0), (40, 0), (40, 1), (40, 6), (40, 15), (40, 16), (40, 39),
                                   (300, 127), (300, 128), (300, 131), (300, 299), (700, 513)])
def test_potrf_not_pd(eng, n, col):
    """first failing column, wherever it falls inside the 4x4 / 16x16 / 128x128 blocking"""
    _lib, H = eng
    A = torch.eye(n, dtype=torch.float64)
    A[col, col] = -1.0
    if col + 3 < n:
        A[col + 3, col + 3] = -2.0                  # a later bad pivot must not win
    Ad = A.cuda()
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(H.lib.gpimhip_potrf(H.h, _lib.ptr(Ad), n, n, _lib.ptr(info)))
    torch.cuda.synchronize()
    assert info.item() == col + 1                   # 1 + first failing column, like LAPACK


@pytest.mark.parametrize("n,noise", [(60, 1e-4), (128, 1e-6), (200, 1e-6), (500, 1e-8)])
def test_potrf_ill_conditioned_kernel_matrix(eng, n, noise):
    """smooth RBF kernel matrix with a small nugget (condition number up to ~1e11): the blocked
    factorisation must stay backward stable -- judged by the residual, not by L itself"""
    _lib, H = eng
    x = torch.linspace(0, 10, n, dtype=torch.float64)
    A = torch.exp(-0.5 * (x[:, None] - x[None, :]) ** 2 / 4.0) + noise * torch.eye(n, dtype=torch.float64)
    Ad = A.cuda().contiguous()
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(H.lib.gpimhip_potrf(H.h, _lib.ptr(Ad), n, n, _lib.ptr(info)))
    torch.cuda.synchronize()
    assert info.item() == 0
    L = torch.tril(Ad.cpu())
    resid = (L @ L.T - A).abs().max().item()
    ref = torch.linalg.cholesky(A)
    resid_ref = (ref @ ref.T - A).abs().max().item()
    assert resid <= max(4 * resid_ref, 1e-14 * n)
    # the log-determinant is a forward quantity: its error is limited by the conditioning (two backward
    # stable factorisations -- LAPACK's and this one -- differ by 1e-13 ... 1e-8 relative on these
    # matrices, tests/tools/illcond_probe.py), so only that much agreement is asked for
    assert_allclose(torch.log(torch.diagonal(L)).sum().item(), torch.log(torch.diagonal(ref)).sum().item(),
                    rtol=1e-7)


CASES = [("RBF", 7, 2, False), ("RBF", 130, 3, False), ("RBF", 300, 2, True),
         ("Matern52", 40, 2, False), ("Matern52", 257, 4, False), ("Matern52", 300, 2, True),
         ("RationalQuadratic", 34, 2, False), ("RationalQuadratic", 200, 3, False),
         # beyond the fused predictor (np > 384: K* slab + variance product on the tile engine) with a ragged last block
         # (GemmArgs::rag): 448 = 3 x 128 + 64 valid rows (the boundary), 530 = 4 x 128 + 18
         ("RBF", 448, 2, False), ("Matern52", 530, 3, False)]


@pytest.mark.parametrize("kind,N,d,iso", CASES)
def test_kmat_nll_grad_predict(eng, kind, N, d, iso):
    _lib, H = eng
    X, y = scattered(N, d, seed=N)
    N = len(X)
    ls = [0.5, 12.0] if iso else [[0.5] * d, [12.0] * d]
    kp, spec, u = pair(kind, d, ls, seed=1)
    m = spec.struct()
    Xd, yd, ud = X.cuda().contiguous(), y.cuda().contiguous(), u.cuda()
    # --- K(X,X) + diag, K(X,Z)
    extra = kp.scale_mixture.detach().reshape(1) if kind == "RationalQuadratic" else torch.ones(1, dtype=torch.float64)
    theta = torch.cat([kp.variance.detach().reshape(1), kp.lengthscale.detach().reshape(-1), extra]).cuda()
    Kd = torch.empty(N, N, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_kmat(H.h, ctypes.byref(m), _lib.ptr(Xd), N, None, 0, _lib.ptr(theta), 0.25,
                                  _lib.ptr(Kd), N))
    Kref = kp.K(X).detach() + 0.25 * torch.eye(N, dtype=torch.float64)
    torch.cuda.synchronize()
    assert_allclose(Kd.cpu().numpy(), Kref.numpy(), rtol=0, atol=1e-13 * kp.variance.item())
    Z = torch.from_numpy(np.random.default_rng(3).uniform(-2, 26, size=(77, d)))
    Zd = Z.cuda().contiguous()
    Kzd = torch.empty(N, 80, dtype=torch.float64, device="cuda")      # ld > M
    _lib.check(H.lib.gpimhip_kmat(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(Zd), 77, _lib.ptr(theta), 0.0,
                                  _lib.ptr(Kzd), 80))
    torch.cuda.synchronize()
    assert_allclose(Kzd.cpu().numpy()[:, :77], kp.K(X, Z).detach().numpy(), rtol=0, atol=1e-13 * kp.variance.item())
    # --- loss and gradient w.r.t. the unconstrained parameters (autograd in the oracle)
    gp = O.ExactGP(X, y, kp, 1e-5)
    loss_ref, g_ref = gp.loss_and_grad()
    out = torch.empty(1 + spec.n_params, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    o = out.cpu()
    assert_allclose(o[0].item(), loss_ref.item(), rtol=1e-12)
    assert_allclose(o[1:].numpy(), g_ref.numpy(), rtol=1e-10, atol=1e-10)
    # --- posterior mean / variance incl. a NaN row
    Xs = torch.from_numpy(np.random.default_rng(5).uniform(0, 24, size=(333, d)))
    Xs[7] = float("nan")
    Xsd = Xs.cuda().contiguous()
    mean = torch.empty(333, dtype=torch.float64, device="cuda")
    var = torch.empty_like(mean)
    _lib.check(H.lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                           _lib.ptr(Xsd), 333, _lib.ptr(mean), _lib.ptr(var)))
    mref, vref = gp.predict(Xs)
    mh, vh = mean.cpu(), var.cpu()
    assert torch.isnan(mh[7]) and torch.isnan(vh[7]) and torch.isnan(mref[7])
    ok = ~torch.isnan(mref)
    assert_allclose(mh[ok].numpy(), mref[ok].numpy(), rtol=0, atol=1e-10)
    assert_allclose(vh[ok].numpy(), vref[ok].numpy(), rtol=0, atol=1e-10)


def test_saturated_interval_has_zero_gradient(eng):
    """u far outside the clipped-sigmoid range: torch.clamp passes no gradient."""
    _lib, H = eng
    X, y = scattered(20, 2, seed=2)
    kp, spec, u = pair("RBF", 2, [[0.5, 0.5], [12., 12.]], seed=3)
    with torch.no_grad():
        kp.u_var.fill_(40.0)
    u[0] = 40.0
    gp = O.ExactGP(X, y, kp, 1e-5)
    _, g_ref = gp.loss_and_grad()
    m = spec.struct()
    Xd, yd, ud = X.cuda().contiguous(), y.cuda().contiguous(), u.cuda()
    out = torch.empty(1 + spec.n_params, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(X), _lib.ptr(ud),
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    g = out.cpu()[1:]
    assert g_ref[0].item() == 0.0 and g[0].item() == 0.0
    assert_allclose(g.numpy(), g_ref.numpy(), rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("kind", ["RBF", "Matern52"])
def test_fit_trajectory(eng, kind):
    """200 Adam iterations: the whole hyper-parameter history follows the oracle."""
    _lib, H = eng
    X, y = scattered(60, 2, seed=3)
    ls = [[0.5, 0.5], [12., 12.]]
    torch.manual_seed(0)
    kp = O.KernelParams(kind, 2, ls)
    gp = O.ExactGP(X, y, kp, 1e-6)
    opt = torch.optim.Adam(kp.parameters(), lr=0.05)
    ref, ref_loss = [], []
    for _ in range(200):
        opt.zero_grad()
        loss = gp.loss()
        loss.backward()
        opt.step()
        ref_loss.append(loss.item())
        ref.append([kp.variance.item(), *kp.lengthscale.tolist(), kp.noise.item()])
    from gpim_amd.kernels import KernelSpec
    torch.manual_seed(0)
    spec = KernelSpec(kind, 2, ls, jitter=1e-6)
    u = spec.draw_initial_u().cuda()
    m = spec.struct()
    Xd, yd = X.cuda().contiguous(), y.cuda().contiguous()
    hist = torch.empty(200, 4, dtype=torch.float64, device="cuda")
    loss = torch.empty(200, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(X), _lib.ptr(u),
                                       0.05, 200, _lib.ptr(hist), _lib.ptr(loss)))
    assert_allclose(hist.cpu().numpy(), np.array(ref), rtol=1e-8)
    assert_allclose(loss.cpu().numpy(), np.array(ref_loss), rtol=1e-10)


def test_fit_reports_not_pd(eng):
    _lib, H = eng
    from gpim_amd.kernels import KernelSpec
    # duplicated points + zero jitter + vanishing noise -> singular K
    X = torch.tensor([[0., 0.], [0., 0.], [1., 1.]], dtype=torch.float64)
    y = torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64)
    torch.manual_seed(0)
    spec = KernelSpec("RBF", 2, [[0.5, 0.5], [2., 2.]], jitter=0.0)
    u = spec.draw_initial_u()
    u[3] = -800.0
    m = spec.struct()
    Xd, yd, ud = X.cuda(), y.cuda(), u.cuda()
    out = torch.empty(5, dtype=torch.float64, device="cuda")
    rc = H.lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), 3, _lib.ptr(ud),
                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8))
    assert rc == _lib.E_NOT_PD
    with pytest.raises(_lib.NotPositiveDefiniteError):
        _lib.check(rc)


@pytest.mark.parametrize("kind,ref", [("cb", None), ("ei", None), ("poi", None)])
def test_acquisition_sweep(eng, kind, ref):
    _lib, H = eng
    rng = np.random.default_rng(11)
    M = 5000
    mean, sd = rng.standard_normal(M), rng.random(M) * 2 + 1e-6
    sd[:50] = 1e-9                                  # |z| huge: saturated Phi
    p0, p1 = (0.3, 1.7) if kind == "cb" else (0.35, 0.01)
    if kind == "cb":
        expect = p0 * mean + p1 * sd
        scale = np.abs(p0 * mean) + np.abs(p1 * sd)
    else:
        imp = mean - p0 - p1
        z = imp / sd
        expect = imp * norm.cdf(z) + sd * norm.pdf(z) if kind == "ei" else norm.cdf(z)
        # EI cancels catastrophically in the lower tail: bound the error by the size of its terms
        # and exp(-z^2/2) carries the rounding of its argument (relative error ~ eps * z^2 / 2)
        scale = np.abs(imp) * norm.cdf(z) + sd * norm.pdf(z) if kind == "ei" else norm.cdf(z)
        scale = scale * (1.0 + 0.02 * z * z)
    md, sdd = torch.from_numpy(mean).cuda(), torch.from_numpy(sd).cuda()
    out = torch.empty(M, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_acq(H.h, _lib.ACQ_IDS[kind], _lib.ptr(md), _lib.ptr(sdd), M, p0, p1, None, _lib.ptr(out)))
    assert np.all(np.abs(out.cpu().numpy() - expect) <= 2e-14 * scale + 1e-300)
    mask = np.ones(M)
    mask[::3] = np.nan
    maskd = torch.from_numpy(mask).cuda()
    _lib.check(H.lib.gpimhip_acq(H.h, _lib.ACQ_IDS[kind], _lib.ptr(md), _lib.ptr(sdd), M, p0, p1, _lib.ptr(maskd),
                                 _lib.ptr(out)))
    got = out.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(mask))
    ok = ~np.isnan(mask)
    assert np.all(np.abs(got[ok] - expect[ok]) <= 2e-14 * scale[ok] + 1e-300)


def test_nanmax_and_topk(eng):
    _lib, H = eng
    rng = np.random.default_rng(4)
    x = rng.standard_normal(70000)
    x[rng.random(70000) < 0.3] = np.nan
    xd = torch.from_numpy(x).cuda()
    out = torch.empty(1, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_nanmax(H.h, _lib.ptr(xd), x.size, _lib.ptr(out)))
    assert out.item() == np.nanmax(x)
    for keep_nan in (0, 1):
        k = 100
        vals = torch.empty(k, dtype=torch.float64, device="cuda")
        idx = torch.empty(k, dtype=torch.int64, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        _lib.check(H.lib.gpimhip_topk(H.h, _lib.ptr(xd), x.size, k, keep_nan, _lib.ptr(vals), _lib.ptr(idx), _lib.ptr(cnt)))
        order = np.argsort(x, kind="stable")          # NaNs last
        if keep_nan:
            expect = order[::-1][:k]
        else:
            expect = order[:np.count_nonzero(~np.isnan(x))][::-1][:k]
        assert cnt.item() == k
        np.testing.assert_array_equal(idx.cpu().numpy(), expect)
        np.testing.assert_array_equal(vals.cpu().numpy(), x[expect])
    # fewer valid values than k
    y = np.full(500, np.nan)
    y[[3, 77, 410]] = [0.5, -1.0, 0.5]
    yd = torch.from_numpy(y).cuda()
    vals = torch.empty(10, dtype=torch.float64, device="cuda")
    idx = torch.empty(10, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(H.lib.gpimhip_topk(H.h, _lib.ptr(yd), 500, 10, 0, _lib.ptr(vals), _lib.ptr(idx), _lib.ptr(cnt)))
    assert cnt.item() == 3
    assert idx.cpu().numpy()[:3].tolist() == [410, 3, 77]      # tie: larger flat index first


def _fit_once(_lib, H, N, T, seed):
    """T Adam iterations on a seeded problem through one handle; returns (history, final u)."""
    from gpim_amd.kernels import KernelSpec
    X, y = scattered(N, 2, seed=seed, grid=64 if N <= 3000 else 128)
    spec = KernelSpec("RBF", 2, [[1., 1.], [20., 20.]], jitter=1e-5)
    u = spec.draw_initial_u(generator=torch.Generator().manual_seed(seed)).cuda()      # private: thread-safe
    m = spec.struct()
    Xd, yd = X.cuda().contiguous(), y.cuda().contiguous()
    hist = torch.empty(T, spec.n_params, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(X), _lib.ptr(u),
                                       0.1, T, _lib.ptr(hist), None))
    return hist.cpu().numpy(), u.cpu().numpy()


def test_handle_reuse_across_sizes_and_regimes(eng):
    """One handle walks through the three regimes (fused small-N trainer, graph-replayed blocked path, plain in-order
    launches at large N) in growing and shrinking order -- workspace, tile plans and the captured graph are re-used or
    rebuilt -- and every result is bit-identical to the one of a fresh handle."""
    _lib, H = eng
    # (1207 / 449 / 448: a ragged last block -- at most 64 valid rows, whose padding the tile engine skips -- between
    # sizes of the same padded order that fill it: what a skipped region holds from the previous fit must never be read)
    sizes = [(100, 12), (700, 10), (1500, 4), (300, 10), (6200, 2), (60, 12), (1500, 4), (6200, 2),
             (1207, 4), (1250, 4), (1207, 4), (1216, 4), (1217, 4), (449, 6), (512, 6), (448, 6)]
    for N, T in sizes:
        got = _fit_once(_lib, H, N, T, seed=N)
        fresh = _lib.Handle()
        try:
            want = _fit_once(_lib, fresh, N, T, seed=N)
        finally:
            fresh.close()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), N


def test_two_handles_interleaved(eng):
    """Handles share no state: alternating calls on two of them give what each gives alone."""
    _lib, H = eng
    H2 = _lib.Handle()
    try:
        a1 = _fit_once(_lib, H, 400, 6, seed=1)
        b1 = _fit_once(_lib, H2, 900, 6, seed=2)
        a2 = _fit_once(_lib, H, 400, 6, seed=1)
        b2 = _fit_once(_lib, H2, 900, 6, seed=2)
    finally:
        H2.close()
    assert np.array_equal(a1[0], a2[0]) and np.array_equal(b1[0], b2[0])


@pytest.mark.parametrize("N,T", [(90, 10), (800, 10), (6200, 2)])
def test_user_stream_matches_default_stream(eng, N, T):
    """A handle created under a non-default torch stream runs on that stream (a fit is ONE in-order stream; mid-size N
    replays a graph captured on the library's capture stream) and returns the same bits as one on the default stream."""
    _lib, H = eng
    want = _fit_once(_lib, H, N, T, seed=7)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        H2 = _lib.Handle()
        try:
            got = _fit_once(_lib, H2, N, T, seed=7)
        finally:
            H2.close()
    s.synchronize()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_concurrent_handles_share_side_streams(eng):
    """The capture stream is per device, shared by all handles (api.hip: SideStreams).  Two threads, each with its own
    handle on its own torch stream, train at the same time at large N (N = 6200, plain launches) and in the
    graph-replayed regime (N = 800): the two models cannot mix up -- results are bit-identical to the sequential ones."""
    import threading
    _lib, H = eng
    jobs = [(6200, 2, 11), (800, 10, 12), (6200, 2, 13), (800, 10, 14)]
    want = [_fit_once(_lib, H, N, T, seed) for N, T, seed in jobs]
    got = [None] * len(jobs)
    errors = []

    def work(i):
        try:
            N, T, seed = jobs[i]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                Hi = _lib.Handle()
                try:
                    got[i] = _fit_once(_lib, Hi, N, T, seed)
                finally:
                    Hi.close()
            s.synchronize()
        except Exception as e:                                   # surfaced in the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for g, w, job in zip(got, want, jobs):
        assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]), job
