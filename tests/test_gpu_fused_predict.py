"""
Fused posterior / acquisition kernel for models with few observations (csrc/predict.hip; SURVEY 8(a) rows a11,
a13: K(X, X*) generated in LDS, never written to HBM) and the single-call acquisition entry point
gpimhip_acquire_exact.

Oracle legs: O.ExactGP.predict and the oracle's acquisition functions (restating gpim/gpreg/gpr.py:243-250 and
gpim/gpbayes/acqfunc.py:11-92); the slab path of the same library (GPIMHIP_NO_FUSED_PREDICT=1), which the
large-N parity tests pin to the oracle, is the second reference.
"""
import ctypes

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(ensure_built):
    from gpim_amd import _lib
    return _lib, _lib.Handle()


def make_problem(N, d, kind, seed):
    """(X, y, oracle KernelParams, KernelSpec, u, extent) with identical parameters on both sides."""
    from gpim_amd.kernels import KernelSpec
    from oracle import gpim_oracle as O
    rng = np.random.default_rng(seed)
    side = int(np.ceil((4 * N) ** (1.0 / d))) + 2
    pts = np.unique(rng.integers(0, side, size=(8 * N, d)), axis=0)
    X = pts[rng.permutation(len(pts))[:N]].astype(np.float64)
    y = np.sin(X.sum(1) / 3.0) + 0.05 * rng.standard_normal(N)
    ls = [[1.0] * d, [float(side)] * d]
    torch.manual_seed(seed)
    kp = O.KernelParams(kind, d, ls)
    torch.manual_seed(seed)
    spec = KernelSpec(kind, d, ls, jitter=1e-5)
    u = spec.draw_initial_u()
    with torch.no_grad():
        kp.u_noise.fill_(-3.0)
    u[1 + spec.n_ls] = -3.0
    return X, y, kp, spec, u, side


def predict(eng, X, y, spec, u, Xs):
    _lib, H = eng
    Xd, yd, ud, Xsd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (X, y, u.numpy(), Xs))
    M = len(Xs)
    mean = torch.empty(M, dtype=torch.float64, device="cuda")
    var = torch.empty_like(mean)
    m = spec.struct()
    _lib.check(H.lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(X), _lib.ptr(ud),
                                           _lib.ptr(Xsd), M, _lib.ptr(mean), _lib.ptr(var)))
    return mean.cpu().numpy(), var.cpu().numpy()


@pytest.mark.parametrize("kind", ["RBF", "Matern52", "RationalQuadratic"])
@pytest.mark.parametrize("N,d,M", [(17, 1, 1), (60, 2, 1000), (128, 2, 33), (129, 3, 257), (300, 2, 4097), (384, 4, 640)])
def test_fused_predict_vs_oracle_and_slab(eng, monkeypatch, kind, N, d, M):
    from oracle import gpim_oracle as O
    X, y, kp, spec, u, side = make_problem(N, d, kind, seed=N + d)
    rng = np.random.default_rng(M)
    Xs = rng.uniform(-1, side + 1, size=(M, d))
    if M > 40:
        Xs[7] = np.nan                                            # NaN rows propagate (gprutils.prepare_test_data)
        Xs[11] = X[3]                                             # a test point on top of an observation
    mean, var = predict(eng, X, y, spec, u, Xs)
    gp = O.ExactGP(torch.from_numpy(X), torch.from_numpy(y), kp, 1e-5)
    mo, vo = (t.numpy() for t in gp.predict(torch.from_numpy(Xs)))
    scale = np.nanmax(np.abs(mo)) + 1.0
    assert_allclose(mean, mo, rtol=0, atol=1e-10 * scale, equal_nan=True)
    assert_allclose(var, vo, rtol=1e-9, atol=1e-10 * np.nanmax(vo), equal_nan=True)
    monkeypatch.setenv("GPIMHIP_NO_FUSED_PREDICT", "1")
    mean_s, var_s = predict(eng, X, y, spec, u, Xs)
    assert_allclose(mean, mean_s, rtol=0, atol=1e-12 * scale, equal_nan=True)
    assert_allclose(var, var_s, rtol=1e-11, atol=1e-12 * np.nanmax(vo), equal_nan=True)


def test_fused_predict_batched_equals_single(eng):
    _lib, H = eng
    N, d, B, M = 90, 2, 5, 700
    X, _, _, spec, u, side = make_problem(N, d, "Matern52", seed=4)
    rng = np.random.default_rng(0)
    Y = np.sin(X.sum(1)[None, :] / (2.0 + np.arange(B)[:, None])) + 0.05 * rng.standard_normal((B, N))
    U = torch.stack([u + 0.1 * i for i in range(B)])
    Xs = rng.uniform(0, side, size=(M, d))
    Xd, Yd, Ud, Xsd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (X, Y, U.numpy(), Xs))
    mean = torch.empty(B, M, dtype=torch.float64, device="cuda")
    var = torch.empty_like(mean)
    m = spec.struct()
    _lib.check(H.lib.gpimhip_predict_exact_batched(H.h, ctypes.byref(m), _lib.ptr(Xd), 0, _lib.ptr(Yd), N, B,
                                                   _lib.ptr(Ud), _lib.ptr(Xsd), M, _lib.ptr(mean), _lib.ptr(var)))
    for b in range(B):
        mb, vb = predict(eng, X, Y[b], spec, U[b], Xs)
        assert np.array_equal(mean[b].cpu().numpy(), mb)
        assert np.array_equal(var[b].cpu().numpy(), vb)


@pytest.mark.parametrize("kind", ["cb", "ei", "poi"])
@pytest.mark.parametrize("N", [40, 200, 500])           # 500: slab path inside the same entry point
@pytest.mark.parametrize("masked", [False, True])
def test_acquire_exact_vs_oracle(eng, kind, N, masked):
    from oracle import gpim_oracle as O
    _lib, H = eng
    d, side_g = 2, 48
    X, y, kp, spec, u, side = make_problem(N, d, "RBF", seed=N)
    ii, jj = np.meshgrid(np.arange(float(side_g)), np.arange(float(side_g)), indexing="ij")
    Xs = np.stack([ii.ravel(), jj.ravel()], 1) * (side / side_g)
    M = len(Xs)
    mask = None
    if masked:
        mask = np.ones(M)
        mask[np.random.default_rng(1).integers(0, M, M // 10)] = np.nan
    p0, p1 = (0.3, 1.7) if kind == "cb" else (0.0, 0.01)
    Xd, yd, ud, Xsd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (X, y, u.numpy(), Xs))
    mean = torch.empty(M, dtype=torch.float64, device="cuda")
    sd, acq = torch.empty_like(mean), torch.empty_like(mean)
    md = None if mask is None else torch.from_numpy(mask).cuda()
    m = spec.struct()
    _lib.check(H.lib.gpimhip_acquire_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                           _lib.ptr(Xsd), M, _lib.ptr(Xd), N, _lib.ACQ_IDS[kind], p0, p1,
                                           None if md is None else _lib.ptr(md), _lib.ptr(mean), _lib.ptr(sd),
                                           _lib.ptr(acq)))
    gp = O.ExactGP(torch.from_numpy(X), torch.from_numpy(y), kp, 1e-5)
    mo, so = (t.numpy() for t in gp.predict(torch.from_numpy(Xs)))
    mobs, sobs = (t.numpy() for t in gp.predict(torch.from_numpy(X)))
    so, sobs = np.sqrt(so), np.sqrt(sobs)
    from scipy.stats import norm
    if kind == "cb":
        ref = p0 * mo + p1 * so
    else:
        best = np.nanmax(mobs) if kind == "ei" else max(np.nanmax(mobs), np.nanmax(sobs))
        imp = mo - best - p1
        z = imp / so
        ref = imp * norm.cdf(z) + so * norm.pdf(z) if kind == "ei" else norm.cdf(z)
    if mask is not None:
        ref = mask * ref
    assert_allclose(mean.cpu().numpy(), mo, rtol=0, atol=1e-10 * (np.abs(mo).max() + 1))
    assert_allclose(sd.cpu().numpy(), so, rtol=1e-8, atol=1e-10)
    # EI / POI amplify the posterior error by 1 / sd (z = imp / sd): tolerance on the scale of the map
    assert_allclose(acq.cpu().numpy(), ref, rtol=1e-6, atol=1e-8 * (np.nanmax(np.abs(ref)) + 1e-300), equal_nan=True)


def test_acquisition_on_device_matches_public_functions(ensure_built):
    """boptimizer's device path (one gpimhip_acquire_exact call) against the public acquisition functions
    (reference call sequence: predict, predict, nanmax, sweep)."""
    import gpim_amd as gpim
    from gpim_amd import acqfunc
    from tests.problems import bo_test_problem
    func, Z = bo_test_problem()
    X_full, X_sparse = gpim.utils.get_full_grid(Z), gpim.utils.get_sparse_grid(Z)
    bo = gpim.boptimizer(X_sparse, Z, X_full, func, acquisition_function="ei", exploration_steps=1, gp_iterations=50,
                         verbose=0)
    sm = bo.surrogate_model
    sm.train()
    for kind, fn in (("cb", acqfunc.confidence_bound), ("ei", acqfunc.expected_improvement),
                     ("poi", acqfunc.probability_of_improvement)):
        acq_d, mean_d, sd_d = acqfunc.acquisition_on_device(sm, kind, X_full, X_sparse, 0.0, 1.0, 0.01)
        args = (sm, X_full) if kind == "cb" else (sm, X_full, X_sparse)
        acq, (mean, sd) = fn(*args, xi=0.01)
        assert_allclose(mean_d.cpu().numpy(), np.asarray(mean).ravel(), rtol=0, atol=1e-12)
        assert_allclose(sd_d.cpu().numpy(), np.asarray(sd).ravel(), rtol=1e-10, atol=1e-13)
        assert_allclose(acq_d.cpu().numpy(), np.asarray(acq).ravel(), rtol=1e-7, atol=1e-12, equal_nan=True)
