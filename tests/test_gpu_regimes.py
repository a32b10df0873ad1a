"""
Oracle parity in every execution regime of the engine, at the sizes the benchmark and the
BASELINE.json configs actually use ("next" item 1):

  * large-N regime (N >= 6144: plain in-order launches -- the step schedule of csrc/cholstep.hip with the inverse riding
    in the factorisation's launches, one stream) -- the path bench.py times: loss / gradient / posterior and three Adam
    iterations against O.ExactGP for Matern52 (N = 8192) and RBF (N = 6400);
  * graph-replayed blocked path (128 < N < 6144): a 200-iteration Adam trajectory at N = 300;
  * general path at BO sizes (GPIMHIP_NO_SMALLN=1): the reference's three golden BO runs;
  * config C5 at its stated size (10 x 10 x 64 x 5, 512 requested inducing points): operator-level
    VFE loss / gradient / posterior against O.SparseGP on one per-Ns slice (N = 6400), a short
    reconstructor(sparse=True) run against the oracle, and reconstruct_slices over the five slices
    bit-equal to stand-alone reconstructors (T = 200).

Tolerances are those of tests/test_gpu_ops.py (loss rel 1e-12, gradient rel 1e-10, posterior abs
1e-10) loosened only where the conditioning of the problem at this size limits a forward quantity
(stated at the assertion).
"""
import ctypes
import os

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

from oracle import gpim_oracle as O
from problems import bo_test_problem, ckpfm_cube, lattice_image, oracle_threads
from test_oracle_golden import ORDER


@pytest.fixture(scope="module")
def gpim(ensure_built):
    import gpim_amd
    return gpim_amd


def _oracle_pair(kind, d, ls, seed, jitter=1e-5):
    from gpim_amd.kernels import KernelSpec
    torch.manual_seed(seed)
    kp = O.KernelParams(kind, d, ls)
    spec = KernelSpec(kind, d, ls, jitter=jitter)
    u = spec.draw_initial_u(torch.Generator().manual_seed(seed))
    return kp, spec, u


def _loss_rtol(make_loss, X, y, loss_ref):
    """The loss is a forward quantity (log-determinant + quadratic form): two backward-stable
    factorisations of the same matrix differ in it by ~cond(K) * eps.  The oracle's own sensitivity to a
    mathematically irrelevant change -- the order of the observations -- measures that for the problem at
    hand: the tolerance is 1e-12 (the small-N bar of tests/test_gpu_ops.py) or 32x that sensitivity,
    whichever is larger, and never more than 1e-10.  (Two permutations are a two-sample estimate of the scale of
    that rounding noise; the factor was 16 until the left-looking step schedule of the Cholesky -- another
    elimination order again -- moved the sparse model's loss at C5's size from 1.9e-12 to 2.06e-12 against
    a sampled sensitivity of 1.26e-13.)"""
    sens = 0.0
    for seed in (1, 2):
        perm = torch.from_numpy(np.random.default_rng(seed).permutation(len(X)))
        sens = max(sens, abs(make_loss(X[perm].contiguous(), y[perm].contiguous()) - loss_ref) / abs(loss_ref))
    return min(max(1e-12, 32.0 * sens), 1e-10), sens


@pytest.mark.parametrize("kind,size,frac", [("Matern52", 128, 0.5), ("RBF", 160, 0.25)])
def test_lookahead_regime_vs_oracle(gpim, kind, size, frac):
    """N = 8192 / 6400: >= 12 outer panels, i.e. the schedule bench.py times at N = 16384."""
    from gpim_amd import _lib
    R, _ = lattice_image(size=size, frac=frac, seed=1)
    X, y = gpim.utils.prepare_training_data(gpim.utils.get_sparse_grid(R), R)
    N = X.shape[0]
    assert N in (8192, 6400) and (N + 127) // 128 >= 48
    ls = [[1., 1.], [20., 20.]]
    kp, spec, u = _oracle_pair(kind, 2, ls, seed=0)
    # a mid-range noise level keeps K's condition number moderate, so forward quantities can be
    # compared at the small-N tolerances
    with torch.no_grad():
        kp.u_noise.fill_(-3.0)
    u[1 + spec.n_ls] = -3.0
    m = spec.struct()
    H = _lib.Handle()
    Xd, yd, ud = X.cuda().contiguous(), y.cuda().contiguous(), u.cuda()
    out = torch.empty(1 + spec.n_params, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    rng = np.random.default_rng(7)
    Xs = torch.from_numpy(rng.uniform(0, size - 1, size=(700, 2)))
    Xs[11] = float("nan")
    Xsd = Xs.cuda().contiguous()
    mean = torch.empty(700, dtype=torch.float64, device="cuda")
    var = torch.empty_like(mean)
    _lib.check(H.lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                           _lib.ptr(Xsd), 700, _lib.ptr(mean), _lib.ptr(var)))
    with oracle_threads():
        gp = O.ExactGP(X, y, kp, 1e-5)
        loss_ref, g_ref = gp.loss_and_grad()
        mref, vref = gp.predict(Xs)
        with torch.no_grad():
            rtol, sens = _loss_rtol(lambda Xp, yp: O.ExactGP(Xp, yp, kp, 1e-5).loss().item(), X, y, loss_ref.item())
    o = out.cpu()
    print("loss rel. diff %.2e (oracle's own order sensitivity %.2e)" % (
        abs(o[0].item() - loss_ref.item()) / abs(loss_ref.item()), sens))
    assert_allclose(o[0].item(), loss_ref.item(), rtol=rtol)
    assert_allclose(o[1:].numpy(), g_ref.numpy(), rtol=1e-10, atol=1e-10 * g_ref.abs().max().item())
    mh, vh = mean.cpu(), var.cpu()
    assert torch.isnan(mh[11]) and torch.isnan(vh[11])
    ok = ~torch.isnan(mref)
    assert_allclose(mh[ok].numpy(), mref[ok].numpy(), rtol=0, atol=1e-10)
    assert_allclose(vh[ok].numpy(), vref[ok].numpy(), rtol=0, atol=1e-10)
    H.close()


def test_lookahead_regime_adam_iterations(gpim):
    """Three Adam iterations + predict through reconstructor at N = 8192 (Matern52), the bench
    workload's kernel, against the oracle: histories rel 1e-9, posterior RMSE < 1e-9."""
    R, _ = lattice_image(size=128, frac=0.5, seed=1)
    X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
    kw = dict(kernel="Matern52", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=3, verbose=0)
    rec = gpim.reconstructor(X, R, Xf, **kw)
    assert rec.X.shape[0] == 8192
    mean, sd, hyper = rec.run()
    with oracle_threads():
        orc = O.reconstructor(X, R, Xf, **kw)
        mo, so, ho = orc.run()
    assert_allclose(hyper["variance"], ho["variance"], rtol=1e-9)
    assert_allclose(hyper["lengthscale"], ho["lengthscale"], rtol=1e-9)
    assert_allclose(hyper["noise"], ho["noise"], rtol=1e-9)
    assert_allclose(rec.loss_all, orc.loss_all, rtol=1e-11)
    assert np.sqrt(np.mean((mean - mo) ** 2)) < 1e-9
    assert np.sqrt(np.mean((sd - so) ** 2)) < 1e-9


@pytest.mark.parametrize("kind", ["RBF", "Matern52"])
def test_fit_trajectory_blocked_graph_path(gpim, kind):
    """200 Adam iterations at N = 300: three 128-blocks, blocked path replayed from a hipGraph
    (finalize_kernel with the device-side iteration counter).  The whole history follows the oracle."""
    from gpim_amd import _lib
    from gpim_amd.kernels import KernelSpec
    rng = np.random.default_rng(11)
    pts = np.unique(rng.integers(0, 40, size=(2000, 2)), axis=0)
    pts = pts[rng.permutation(len(pts))[:300]]
    X = torch.from_numpy(pts.astype(np.float64))
    y = torch.from_numpy(np.sin(pts[:, 0] / 5.0) * np.cos(pts[:, 1] / 7.0) + 0.05 * rng.standard_normal(300))
    ls = [[0.5, 0.5], [20., 20.]]
    T = 200
    torch.manual_seed(0)
    kp = O.KernelParams(kind, 2, ls)
    gp = O.ExactGP(X, y, kp, 1e-5)
    opt = torch.optim.Adam(kp.parameters(), lr=0.05)
    ref, ref_loss = [], []
    with oracle_threads(8):
        for _ in range(T):
            opt.zero_grad()
            loss = gp.loss()
            loss.backward()
            opt.step()
            ref_loss.append(loss.item())
            ref.append([kp.variance.item(), *kp.lengthscale.tolist(), kp.noise.item()])
    spec = KernelSpec(kind, 2, ls, jitter=1e-5)
    u = spec.draw_initial_u(torch.Generator().manual_seed(0)).cuda()
    m = spec.struct()
    H = _lib.Handle()
    Xd, yd = X.cuda().contiguous(), y.cuda().contiguous()
    hist = torch.empty(T, 4, dtype=torch.float64, device="cuda")
    loss = torch.empty(T, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), 300, _lib.ptr(u),
                                       0.05, T, _lib.ptr(hist), _lib.ptr(loss)))
    assert_allclose(hist.cpu().numpy(), np.array(ref), rtol=1e-8)
    assert_allclose(loss.cpu().numpy(), np.array(ref_loss), rtol=1e-10)
    H.close()


@pytest.mark.parametrize("acqf", ["ei", "poi", "cb"])
def test_boptim_golden_general_path(gpim, acqf, golden_dir, tmp_path, monkeypatch):
    """The reference's golden BO runs (test/test_boptim.py:42-58) with the fused small-N trainer
    switched off: 21 trainings x 1000 iterations through kmat -> potf2 -> trtri -> lauum ->
    grad_reduce -> finalize, graph-replayed.  Same golden vector, same query order."""
    monkeypatch.setenv("GPIMHIP_NO_SMALLN", "1")
    trial_func, Z_sparse = bo_test_problem()
    bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z_sparse), Z_sparse, gpim.utils.get_full_grid(Z_sparse),
                         trial_func, acquisition_function=acqf, exploration_steps=20, use_gpu=False, verbose=0,
                         filename=str(tmp_path / "bo"))
    bo.run()
    expected = np.load(os.path.join(golden_dir, "test_%s.npy" % acqf))
    assert_allclose(bo.target_func_vals[-1], expected)
    assert [tuple(i) for i in bo.indices_all] == ORDER[acqf]


# ---------------------------------------------------------------------------------------------
# config C5 at its stated size
# ---------------------------------------------------------------------------------------------
def test_c5_full_size_operator_level(gpim):
    """One per-Ns slice of the 10 x 10 x 64 x 5 twin: N = 6400 points in 3-D, indpoints=512 ->
    Xu = X[::12] (534 inducing inputs, gpr.py:151), RBF."""
    from gpim_amd import _lib
    cube = ckpfm_cube()
    assert cube.shape == (10, 10, 64, 5)
    R = cube[..., 2]
    Xg = gpim.utils.get_full_grid(R)
    X, y = gpim.utils.prepare_training_data(Xg, R)
    N, d = X.shape
    assert (N, d) == (6400, 3)
    ls = [[0., 0., 0.], [14., 14., 14.]]          # the reference's default: mean(R.shape) / 2 = 14
    kp, spec, u_t = _oracle_pair("RBF", 3, ls, seed=0)
    with torch.no_grad():
        kp.u_noise.fill_(-2.0)
    u_t[1 + spec.n_ls] = -2.0
    Xu0 = X[::N // 512].clone()
    Mu = len(Xu0)
    assert Mu == 534
    u = torch.cat([u_t, Xu0.reshape(-1)]).cuda()
    m = spec.struct()
    H = _lib.Handle()
    Xd, yd = X.cuda().contiguous(), y.cuda().contiguous()
    P = spec.n_params
    out = torch.empty(1 + P + Mu * d, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_vfe_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(u),
                                          ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    rng = np.random.default_rng(9)
    Xs = torch.from_numpy(rng.uniform(0, 1, size=(600, 3)) * np.array([9., 9., 63.]))
    Xsd = Xs.cuda().contiguous()
    mean = torch.empty(600, dtype=torch.float64, device="cuda")
    var = torch.empty_like(mean)
    _lib.check(H.lib.gpimhip_predict_vfe(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(u),
                                         _lib.ptr(Xsd), 600, _lib.ptr(mean), _lib.ptr(var)))
    with oracle_threads():
        gp = O.SparseGP(X, y, kp, Xu0, 1e-5)
        loss_ref, g_ref = gp.loss_and_grad()
        mr, vr = gp.predict(Xs)
        with torch.no_grad():
            rtol, sens = _loss_rtol(lambda Xp, yp: O.SparseGP(Xp, yp, kp, Xu0, 1e-5).loss().item(), X, y,
                                    loss_ref.item())
    o = out.cpu()
    print("loss rel. diff %.2e (oracle's own order sensitivity %.2e)" % (
        abs(o[0].item() - loss_ref.item()) / abs(loss_ref.item()), sens))
    assert_allclose(o[0].item(), loss_ref.item(), rtol=rtol)
    assert_allclose(o[1:1 + P].numpy(), g_ref[:P].numpy(), rtol=1e-9, atol=1e-9)
    # d loss / d Xu runs through the factorisation of Kuu (534 inducing inputs 12 grid points apart under
    # a lengthscale of ~7: cond(Kuu + jitter I) ~ 1e6), so its rounding error is ~cond * eps, not eps.  The
    # oracle's own sensitivity to the ORDER of the inducing inputs (a different elimination order of the
    # same matrix) measures it; tolerance = the small-N bar (1e-9) or 16x that sensitivity, at most 1e-7.
    with oracle_threads():
        perm = torch.from_numpy(np.random.default_rng(1).permutation(Mu))
        _, g_perm = O.SparseGP(X, y, kp, Xu0[perm].clone(), 1e-5).loss_and_grad()
    gx_ref = g_ref[P:].reshape(Mu, d)
    sens_g = (g_perm[P:].reshape(Mu, d) - gx_ref[perm]).abs().max().item()
    atol_g = min(max(1e-9 * max(1.0, gx_ref.abs().max().item()), 16.0 * sens_g), 1e-7)
    print("max |d grad_Xu| %.2e (oracle's own order sensitivity %.2e)" % (
        (o[1 + P:].reshape(Mu, d) - gx_ref).abs().max().item(), sens_g))
    assert_allclose(o[1 + P:].numpy(), g_ref[P:].numpy(), rtol=0, atol=atol_g)
    assert_allclose(mean.cpu().numpy(), mr.numpy(), atol=1e-10)
    assert_allclose(var.cpu().numpy(), vr.numpy(), atol=1e-10)
    H.close()


def test_c5_full_size_slices(gpim):
    """reconstruct_slices over the five Ns slices (T = 200, as SURVEY 8(d) config 5) is bit-equal to
    stand-alone reconstructor(sparse=True) runs; a 25-iteration run of one slice follows the oracle
    (drift analysis: tests/tools/c5_probe.py)."""
    from gpim_amd import dist as gd
    cube = ckpfm_cube()
    kw = dict(kernel="RBF", sparse=True, indpoints=512, learning_rate=0.05, iterations=200)
    mean, sd, hyper = gd.reconstruct_slices(cube, axis=-1, return_hyperparams=True, **kw)
    assert mean.shape == cube.shape and np.isfinite(mean).all() and np.isfinite(sd).all()
    assert len(hyper) == 5 and len(hyper[0]["noise"]) == 200
    # the fit explains the data: residual RMS at the noise level of the twin (0.02), far below its range
    assert np.sqrt(np.mean((mean - cube) ** 2)) < 0.05
    for k in (0, 4):
        R = cube[..., k]
        Xf = gpim.utils.get_full_grid(R)
        m1, s1, h1 = gpim.reconstructor(Xf, R, Xf, verbose=0, **kw).run()
        np.testing.assert_array_equal(mean[..., k], m1)
        np.testing.assert_array_equal(sd[..., k], s1)
        np.testing.assert_array_equal(np.array(hyper[k]["noise"]), np.array(h1["noise"]))
    R = cube[..., 1]
    Xf = gpim.utils.get_full_grid(R)
    T = 25
    kw25 = dict(kw, iterations=T)
    rec = gpim.reconstructor(Xf, R, Xf, verbose=0, **kw25)
    m25, s25, h25 = rec.run()
    # the oracle's training loop (O.reconstructor.train) unrolled, to record the gradient magnitudes
    with oracle_threads():
        orc = O.reconstructor(Xf, R, Xf, verbose=0, **kw25)
        opt = torch.optim.Adam(orc.model.parameters(), lr=0.05)
        gmin, hist = None, {"variance": [], "lengthscale": [], "noise": []}
        for _ in range(T):
            opt.zero_grad()
            orc.model.loss().backward()
            g = orc.model.Xu.grad.abs().clone()
            gmin = g if gmin is None else torch.minimum(gmin, g)
            opt.step()
            hist["variance"].append(orc.kernel.variance.item())
            hist["lengthscale"].append(orc.kernel.lengthscale.tolist())
            hist["noise"].append(orc.kernel.noise.item())
        xu_ref = orc.model.Xu.detach().numpy()
        mo, so = orc.predict()
    assert_allclose(h25["variance"], hist["variance"], rtol=1e-9)
    assert_allclose(h25["lengthscale"], hist["lengthscale"], rtol=1e-9)
    assert_allclose(h25["noise"], hist["noise"], rtol=1e-9)
    assert_allclose(m25, mo, atol=1e-8)
    assert_allclose(s25, so, atol=1e-8)
    # Inducing inputs: with the noise still near its initial value of 1 the bound is flat in most
    # inducing coordinates (|d loss / d Xu| ~ 1e-9 ... 1e-6 here), and Adam moves every coordinate by
    # ~lr * g / (|g| + 1e-8) per step whatever |g| is: an absolute gradient difference of 1e-10 -- an order
    # of magnitude below the gradient tolerance of the operator-level test -- becomes a 1e-4 difference
    # in position per step where |g| ~ 1e-8.  Hence: tight agreement where the gradient is well above
    # that noise for the whole run, and a random-walk bound (a fraction of lr per step) elsewhere.
    d = np.abs(h25["inducing_points"][-1] - xu_ref)
    well = gmin.numpy() > 1e-4
    assert well.sum() >= 20
    assert d[well].max() < 1e-5, d[well].max()
    assert d.max() < 0.1 * 0.05 * T, d.max()


# ---------------------------------------------------------------------------------------------
# non-positive-definite covariance during training
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("npts,small", [(40, True), (40, False), (300, False)])
def test_not_pd_freezes_at_failing_iteration(gpim, npts, small, monkeypatch):
    """Duplicated inputs with identical targets, zero jitter and a large learning rate drive the noise
    to zero until K stops being numerically positive-definite.  The reference raises inside the loop
    (gpr.py:192) with the history up to the failing iteration; here the device loop freezes there:
    the exception class is the same, the history has one row per completed iteration and follows the
    oracle's, the parameters stay finite, and the reconstructor can be used again."""
    if not small:
        monkeypatch.setenv("GPIMHIP_NO_SMALLN", "1")
    rng = np.random.default_rng(4)
    side = int(np.ceil(np.sqrt(npts)))
    R = np.sin(np.arange(side)[:, None] / 3.0) * np.cos(np.arange(side)[None, :] / 4.0)
    Xg = gpim.utils.get_full_grid(R).astype(np.float64)
    # every point twice: exactly singular K without noise
    Xd = np.concatenate([Xg, Xg], axis=2)
    Rd = np.concatenate([R, R], axis=1)
    kw = dict(kernel="RBF", lengthscale=[[1., 1.], [10., 10.]], learning_rate=1.0, iterations=400, verbose=0,
              jitter=0.0)
    rec = gpim.reconstructor(Xd, Rd, Xd, **kw)
    u_before = rec._u.clone()
    with pytest.raises(torch.linalg.LinAlgError):
        rec.train()
    n_done = len(rec.hyperparams["noise"])
    assert 0 < n_done < 400
    assert torch.isfinite(rec._u).all() and not torch.equal(rec._u.cpu(), u_before.cpu())
    orc = O.reconstructor(Xd, Rd, Xd, **kw)
    with pytest.raises(torch.linalg.LinAlgError):
        orc.train()
    n_ref = len(orc.hyperparams["noise"])
    # borderline pivots: the two factorisations may give up an iteration or two apart
    # (a duplicated-point K is exactly singular without the noise term; a Cholesky survives until
    # noise ~ N * eps * |K| ~ 1e-12 ... 1e-11 here, give or take a decade -- i.e. a few lr = 1 steps --
    # depending on the summation order of the factorisation)
    assert abs(n_done - n_ref) <= 8, (n_done, n_ref)
    # (the last iterations before the failure run on a numerically singular K: compare up to there)
    k = min(n_done, n_ref) - 10
    assert_allclose(rec.hyperparams["noise"][:k], orc.hyperparams["noise"][:k], rtol=1e-4)
    assert_allclose(rec.hyperparams["lengthscale"][:k], orc.hyperparams["lengthscale"][:k], rtol=1e-4)
    # still usable: with a jitter the frozen parameters give a finite posterior
    rec._mstruct.jitter = 1e-6
    mean, sd = rec.predict()
    assert np.isfinite(mean).all() and np.isfinite(sd).all()


def _large_fit(gpim, N, T, precision, seed=0):
    """T Adam iterations of gpimhip_fit_exact in the large-N regime on a fresh handle; (history, losses, final u)."""
    from gpim_amd import _lib
    from gpim_amd.kernels import KernelSpec
    dev = torch.device("cuda:0")
    side = int(np.ceil(np.sqrt(N * 4)))
    rng = np.random.default_rng(seed)
    flat = np.sort(rng.choice(side * side, size=N, replace=False))
    X = np.stack([flat // side, flat % side], 1).astype(np.float64)
    y = np.sin(X[:, 0] / 7.0) * np.cos(X[:, 1] / 5.0) + 0.05 * rng.standard_normal(N)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5)
    u = spec.draw_initial_u(torch.Generator().manual_seed(seed)).to(dev)
    m = spec.struct()
    H = _lib.Handle(precision=precision)
    hist = torch.zeros(T, spec.n_params, dtype=torch.float64, device=dev)
    loss = torch.zeros(T, dtype=torch.float64, device=dev)
    _lib.check(H.lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, T,
                                       _lib.ptr(hist), _lib.ptr(loss)))
    torch.cuda.synchronize()
    out = (hist.cpu().numpy().copy(), loss.cpu().numpy().copy(), u.cpu().numpy().copy())
    H.close()
    return out


@pytest.mark.parametrize("precision", ["double", "single"])
def test_large_n_eager_regime_reproducible(gpim, precision):
    """N = 6200 is just inside the large-N regime (np = 6272, 13 panels of 512 columns >= EAGER_MIN_PANELS = 12): the
    iteration is enqueued launch by launch on the caller's stream (no captured graph, no helper stream).  Two fits
    from the same draw on fresh handles give the same bits -- the launch plan, the hosted tile lists and every
    reduction order are functions of the problem size alone."""
    a = _large_fit(gpim, 6200, 9, precision)
    b = _large_fit(gpim, 6200, 9, precision)
    assert np.isfinite(a[0]).all() and np.isfinite(a[1]).all()
    for x, y_ in zip(a, b):
        assert np.array_equal(x, y_)


def test_graph_replay_equals_eager_launches(gpim, monkeypatch):
    """Mid-size N: one captured iteration replayed T times (default) against the same launches enqueued iteration by
    iteration (GPIMHIP_NO_GRAPH=1, what the profiling tools use) -- the same bits, exact GP at N = 700 (6 block columns:
    the triangular inverse rides in the factorisation's launches in both)."""
    monkeypatch.delenv("GPIMHIP_NO_GRAPH", raising=False)
    a = _large_fit(gpim, 700, 12, "double")
    monkeypatch.setenv("GPIMHIP_NO_GRAPH", "1")
    b = _large_fit(gpim, 700, 12, "double")
    assert np.isfinite(a[0]).all() and np.isfinite(a[1]).all()
    for x, y_ in zip(a, b):
        assert np.array_equal(x, y_)


@pytest.mark.parametrize("n,T,kernel,prec", [(300, 40, "RBF", "double"), (700, 12, "Matern52", "double"),
                                             (2100, 8, "RationalQuadratic", "double"), (8300, 3, "Matern52", "double"),
                                             (700, 8, "RBF", "single")])
def test_fused_finalize_equals_two_launches(gpim, monkeypatch, n, T, kernel, prec):
    """Round 6: the finalize step (loss, chain rule, Adam step, history row, the NEXT iteration's theta) runs in the last
    workgroup of the gradient-contraction launch, alpha reaches that launch as the row-chunk partial sums of the triangular
    mat-vec (np <= 8192), and no theta launch remains inside the loop -- against the separate launches of rounds 1-5
    (GPIMHIP_NO_FUSED_FINALIZE=1): the same reductions in the same order, hence the same bits in the hyper-parameter history,
    the loss history and the posterior; sizes on both sides of the graph-replay / eager switch and of the partial-sums
    switch; a second run must repeat the first (the partial sums cross XCDs through device-scope stores and loads: a stale
    read would show here)."""
    side = int(np.ceil(np.sqrt(n * 4)))
    rng = np.random.default_rng(n)
    ii, jj = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    R = np.sin(ii / 7.0) * np.cos(jj / 5.0) + 0.05 * rng.standard_normal((side, side))
    R.ravel()[rng.permutation(side * side)[n:]] = np.nan            # exactly n observations
    X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
    kw = dict(kernel=kernel, lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=T, verbose=0, precision=prec)
    outs = []
    for knob in (None, None, "1"):
        if knob:
            monkeypatch.setenv("GPIMHIP_NO_FUSED_FINALIZE", knob)
        else:
            monkeypatch.delenv("GPIMHIP_NO_FUSED_FINALIZE", raising=False)
        rec = gpim.reconstructor(X, R, Xf, **kw)
        mean, sd, hyper = rec.run()
        outs.append((mean, sd, np.asarray(hyper["lengthscale"]), np.asarray(hyper["noise"]), np.asarray(hyper["variance"]),
                     np.asarray(rec.loss_all)))
    assert int(np.isfinite(R).sum()) == n and np.isfinite(outs[0][0]).all()
    for a, b in ((outs[0], outs[1]), (outs[0], outs[2])):
        for x, y_ in zip(a, b):
            assert np.array_equal(x, y_)
