"""
Single-precision engine (gpimhip_set_precision(h, 32); reconstructor(precision='single'), gpim/gpreg/gpr.py:104-113):
float N x N matrices, O(N^3) products on v_mfma_f32_16x16x4_f32, diagonal blocks / vectors / loss / gradient / Adam
in double.

Two references: the fp64 oracle (O.ExactGP, the truth) and a plain float32 torch-CPU evaluation of the same
formulas (what a float32 run of the reference computes: float32 covariance, float32 torch.linalg.cholesky,
float32 triangular solves).  The bar: close to the truth at float32 round-off level, and never further from it
than a few times the float32 torch run.
"""
import ctypes
import math

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

from oracle import gpim_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(ensure_built):
    from gpim_amd import _lib
    return _lib, _lib.Handle(precision="single"), _lib.Handle()


def problem(N, d, kind, seed):
    from gpim_amd.kernels import KernelSpec
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
        kp.u_noise.fill_(-2.0)
    u[1 + spec.n_ls] = -2.0
    Xs = rng.uniform(0, side, size=(500, d))
    return X, y, kp, spec, u, Xs


def run_engine(_lib, H, X, y, spec, u, Xs):
    Xd, yd, ud, Xsd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (X, y, u.numpy(), Xs))
    m = spec.struct()
    out = torch.empty(1 + spec.n_params, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(X), _lib.ptr(ud),
                                      ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    M = len(Xs)
    mean = torch.empty(M, dtype=torch.float64, device="cuda")
    var = torch.empty_like(mean)
    _lib.check(H.lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(X), _lib.ptr(ud),
                                           _lib.ptr(Xsd), M, _lib.ptr(mean), _lib.ptr(var)))
    o = out.cpu().numpy()
    return o[0], o[1:], mean.cpu().numpy(), var.cpu().numpy()


def torch_float32_run(kp, X, y, Xs, jitter):
    """loss and posterior as a float32 run computes them (covariance rounded to float32, float32 LAPACK)."""
    Xt, yt, Xst = torch.from_numpy(X), torch.from_numpy(y), torch.from_numpy(Xs)
    with torch.no_grad():
        K = kp.K(Xt).float()
        N = K.shape[0]
        K.view(-1)[::N + 1] += float(jitter + kp.noise)
        L = torch.linalg.cholesky(K)
        z = torch.linalg.solve_triangular(L, yt.float().unsqueeze(-1), upper=False).squeeze(-1)
        loss = 0.5 * (z * z).sum() + L.diagonal().log().sum() + 0.5 * N * math.log(2 * math.pi) + float(kp.neg_log_prior())
        Ks = kp.K(Xt, Xst).float()
        W = torch.linalg.solve_triangular(L, Ks, upper=False)
        mean = W.t() @ z
        var = (kp.Kdiag(Xst).float() - (W * W).sum(0)).clamp(min=0) + float(kp.noise)
    return float(loss), mean.double().numpy(), var.double().numpy()


@pytest.mark.parametrize("kind,N,d,schedule", [("RBF", 300, 2, "steps"), ("Matern52", 700, 2, "steps"),
                                               ("RationalQuadratic", 260, 3, "steps"), ("Matern52", 1500, 2, "steps"),
                                               ("RBF", 2300, 2, "steps")])
def test_single_engine_vs_truth_and_float32_run(eng, kind, N, d, schedule):
    """The float step schedule of csrc/cholstep32.hip."""
    _lib, H32, H64 = eng
    X, y, kp, spec, u, Xs = problem(N, d, kind, seed=N)
    gp = O.ExactGP(torch.from_numpy(X), torch.from_numpy(y), kp, 1e-5)
    loss_t, grad_t = gp.loss_and_grad()
    loss_t, grad_t = loss_t.item(), grad_t.numpy()
    mean_t, var_t = (t.numpy() for t in gp.predict(torch.from_numpy(Xs)))
    loss, grad, mean, var = run_engine(_lib, H32, X, y, spec, u, Xs)
    # float32 round-off level (times the conditioning of these covariances) against the truth
    assert_allclose(loss, loss_t, rtol=5e-5)
    assert_allclose(grad, grad_t, rtol=0, atol=5e-4 * np.abs(grad_t).max())
    assert_allclose(mean, mean_t, rtol=0, atol=2e-3 * (np.abs(mean_t).max() + 1))
    assert_allclose(var, var_t, rtol=2e-2, atol=1e-4)
    # against a float32 run of the same formulas: the posterior mean (alpha is refined against the covariance in
    # double) is at least as close to the truth; the variance and the loss (fp32 factor, explicit fp32 inverse)
    # stay within a small multiple / at 2e-5 of the loss
    loss_f, mean_f, var_f = torch_float32_run(kp, X, y, Xs, 1e-5)
    assert np.abs(mean - mean_t).max() <= np.abs(mean_f - mean_t).max() + 1e-6
    assert np.abs(var - var_t).max() <= 8 * np.abs(var_f - var_t).max() + 1e-6
    assert abs(loss - loss_t) <= max(4 * abs(loss_f - loss_t), 2e-5 * abs(loss_t))
    # the double-precision handle is untouched by its neighbour
    loss64, grad64, mean64, _ = run_engine(_lib, H64, X, y, spec, u, Xs)
    assert_allclose(loss64, loss_t, rtol=1e-11)
    assert_allclose(mean64, mean_t, rtol=0, atol=1e-9)


def test_precision_switch_and_guards(ensure_built):
    from gpim_amd import _lib
    X, y, kp, spec, u, Xs = problem(400, 2, "RBF", seed=1)
    Ha, Hb = _lib.Handle(), _lib.Handle()
    ref = run_engine(_lib, Ha, X, y, spec, u, Xs)
    _lib.check(Hb.lib.gpimhip_set_precision(Hb.h, 32))
    single = run_engine(_lib, Hb, X, y, spec, u, Xs)
    assert single[0] != ref[0] and abs(single[0] - ref[0]) < 1e-5 * abs(ref[0])
    # entry points that keep their matrices in double refuse a single-precision handle
    A = torch.eye(256, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = Hb.lib.gpimhip_potrf(Hb.h, _lib.ptr(A), 256, 256, _lib.ptr(info))
    assert rc == _lib.E_BADARG and "double-precision handle" in Hb.lib.gpimhip_last_error().decode()
    # back to double: bit-identical to a handle that never switched
    _lib.check(Hb.lib.gpimhip_set_precision(Hb.h, 64))
    again = run_engine(_lib, Hb, X, y, spec, u, Xs)
    assert again[0] == ref[0] and np.array_equal(again[1], ref[1]) and np.array_equal(again[2], ref[2])
    assert Hb.lib.gpimhip_set_precision(Hb.h, 16) == _lib.E_BADARG


def test_reconstructor_single_precision(ensure_built):
    """precision='single' end to end: float32 results, hyper-parameter trajectory and posterior close to the
    double-precision run (general path: N > 128), half the workspace."""
    import gpim_amd as gpim
    from tests.problems import spiral_image
    R, _ = spiral_image(size=48, keep=0.3, seed=2)
    X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
    kw = dict(kernel="Matern52", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=40, verbose=0)
    rd = gpim.reconstructor(X, R, Xf, **kw)
    rs = gpim.reconstructor(X.astype(np.float32), R.astype(np.float32), Xf.astype(np.float32), precision="single", **kw)
    # like the reference, precision='single' draws the initial hyper-parameters in float32 (different numbers from
    # the same seed); start both runs from the same point to compare the arithmetic
    rs._u.copy_(rd._u)
    md, sdd, hd = rd.run()
    ms, sds, hs = rs.run()
    assert ms.dtype == np.float32 and sds.dtype == np.float32
    assert rs._handle.precision == "single" and rd._handle.precision == "double"
    assert_allclose(ms, md, rtol=0, atol=2e-3)
    assert_allclose(sds, sdd, rtol=2e-2, atol=1e-3)
    assert_allclose(np.asarray(hs["noise"], dtype=np.float64), np.asarray(hd["noise"], dtype=np.float64), rtol=2e-2)
    wd = rd._handle.lib.gpimhip_workspace_bytes(rd._handle.h)
    ws = rs._handle.lib.gpimhip_workspace_bytes(rs._handle.h)
    assert ws < 0.62 * wd
    # sparse models keep the double-precision engine whatever the precision flag says
    rsp = gpim.reconstructor(X, R, Xf, sparse=True, indpoints=40, precision="single", **kw)
    assert rsp._handle.precision == "double"


def test_single_precision_not_positive_definite_is_reported(eng):
    """A covariance that float32 cannot factor (duplicate points, no noise to speak of) is reported like the
    reference's float32 run reports it: linalg.cholesky error, not garbage."""
    _lib, H32, _ = eng
    from gpim_amd.kernels import KernelSpec
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 4, size=(300, 2))
    X[150:] = X[:150] + 1e-7                                   # near-duplicate rows
    y = np.sin(X.sum(1))
    spec = KernelSpec("RBF", 2, [[1., 1.], [20., 20.]], jitter=1e-12)
    torch.manual_seed(0)
    u = spec.draw_initial_u()
    u[1 + spec.n_ls] = -40.0                                   # noise = exp(-40)
    Xd, yd, ud = (torch.from_numpy(a).cuda() for a in (X, y, u.numpy()))
    m = spec.struct()
    out = torch.empty(1 + spec.n_params, dtype=torch.float64, device="cuda")
    rc = H32.lib.gpimhip_nll_grad(H32.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(X), _lib.ptr(ud),
                                  ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8))
    assert rc == _lib.E_NOT_PD


@pytest.mark.parametrize("N", [150, 700])
def test_acquire_exact_on_single_precision_handle(eng, N):
    """The one-call acquisition entry point on a single-precision handle (slab path: the fused small-N kernel reads
    L^-1 in double) against the double-precision handle."""
    _lib, H32, H64 = eng
    X, y, kp, spec, u, _ = problem(N, 2, "Matern52", seed=N + 5)
    side = X.max() + 1
    g = np.stack(np.meshgrid(np.linspace(0, side, 40), np.linspace(0, side, 40), indexing="ij"), -1).reshape(-1, 2)
    out = {}
    for name, H in (("s", H32), ("d", H64)):
        Xd, yd, ud, gd = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (X, y, u.numpy(), g))
        M = len(g)
        mean = torch.empty(M, dtype=torch.float64, device="cuda")
        sd, acq = torch.empty_like(mean), torch.empty_like(mean)
        m = spec.struct()
        _lib.check(H.lib.gpimhip_acquire_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                               _lib.ptr(gd), M, _lib.ptr(Xd), N, _lib.ACQ_IDS["ei"], 0.0, 0.01, None,
                                               _lib.ptr(mean), _lib.ptr(sd), _lib.ptr(acq)))
        out[name] = (mean.cpu().numpy(), sd.cpu().numpy(), acq.cpu().numpy())
    assert_allclose(out["s"][0], out["d"][0], rtol=0, atol=2e-4)
    assert_allclose(out["s"][1], out["d"][1], rtol=2e-3, atol=1e-5)
    assert_allclose(out["s"][2], out["d"][2], rtol=0, atol=2e-3 * (np.abs(out["d"][2]).max() + 1e-12))


def test_slices_and_bo_in_single_precision(ensure_built, tmp_path):
    import gpim_amd as gpim
    from gpim_amd import dist as gdist
    from tests.problems import hyperspectral_cube, bo_test_problem
    cube, _ = hyperspectral_cube(size=24, nspec=3, keep=0.4, seed=1)
    kw = dict(kernel="RBF", lengthscale=[[1., 1.], [12., 12.]], learning_rate=0.1, iterations=30)
    md, sd_ = gdist.reconstruct_slices(cube, axis=-1, batch=3, **kw)
    ms, ss = gdist.reconstruct_slices(cube.astype(np.float32), axis=-1, batch=3, precision="single", **kw)
    assert np.isfinite(ms).all() and np.isfinite(ss).all()
    # float32 initial draws differ from the float64 ones (as in the reference): same model family, nearby optimum
    assert np.abs(ms - md).max() < 0.2 and np.abs(ss - sd_).max() < 0.2
    func, Z = bo_test_problem()
    bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), func, acquisition_function="ei",
                         exploration_steps=4, gp_iterations=100, verbose=0, precision="single",
                         filename=str(tmp_path / "bo"))
    bo.run()
    assert len(bo.indices_all) == 4 and all(0 <= i < 25 and 0 <= j < 25 for i, j in bo.indices_all)
    mean, sd = bo.gp_predictions[-1]
    assert np.asarray(mean).dtype == np.float32 and np.isfinite(np.asarray(mean)).all()
