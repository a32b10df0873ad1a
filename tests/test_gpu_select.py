"""
Ranking kernels for large candidate grids (csrc/select.hip) and the device-resident acquisition path of
boptimizer: multi-block radix top-k and two-stage nanmax against numpy (the reference ranks with
np.argsort(...)[::-1], boptim.py:303-315), batch thinning against the reference's own update_points
outputs (tests/golden/host_logic.npz), and a BO run on a 256 x 256 grid.
"""
import ctypes
import os

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(ensure_built):
    from gpim_amd import _lib
    return _lib, _lib.Handle()


def ref_topk(x, k, keep_nan):
    """Descending by (value, flat index); NaN first when kept (np.argsort puts NaN last, the caller reverses);
    ties: larger index first (= reversed stable sort)."""
    idx = np.arange(len(x))
    if keep_nan:
        key = np.where(np.isnan(x), np.inf, x)
        nanflag = np.isnan(x).astype(int)
        order = np.lexsort((idx, key, nanflag))[::-1]
    else:
        ok = ~np.isnan(x)
        order = idx[ok][np.lexsort((idx[ok], x[ok]))[::-1]]
    return order[:k]


def run_topk(eng, x, k, keep_nan):
    _lib, H = eng
    xd = torch.from_numpy(x).cuda()
    vals = torch.empty(k, dtype=torch.float64, device="cuda")
    idx = torch.empty(k, dtype=torch.int64, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    _lib.check(H.lib.gpimhip_topk(H.h, _lib.ptr(xd), len(x), k, keep_nan, _lib.ptr(vals), _lib.ptr(idx), _lib.ptr(cnt)))
    n = int(cnt.item())
    return vals.cpu().numpy()[:n], idx.cpu().numpy()[:n], vals.cpu().numpy()[n:], idx.cpu().numpy()[n:]


@pytest.mark.parametrize("M", [2049, 70001, 1 << 20])
@pytest.mark.parametrize("k", [1, 100, 1000])
@pytest.mark.parametrize("keep_nan", [0, 1])
def test_radix_topk_vs_numpy(eng, M, k, keep_nan):
    rng = np.random.default_rng(M + k)
    x = rng.standard_normal(M)
    x[rng.integers(0, M, M // 50)] = np.nan                  # NaNs
    x[rng.integers(0, M, M // 20)] = 1.25                    # many exact ties, some at the top
    x[rng.integers(0, M, 30)] = np.inf
    x[rng.integers(0, M, 30)] = -np.inf
    x[rng.integers(0, M, 7)] = 0.0
    x[rng.integers(0, M, 7)] = -0.0
    v, i, vpad, ipad = run_topk(eng, x, k, keep_nan)
    want = ref_topk(x, k, keep_nan)
    assert len(i) == len(want)
    np.testing.assert_array_equal(i, want)
    np.testing.assert_array_equal(v, x[want])
    assert np.isnan(vpad).all() and (ipad == -1).all()


def test_radix_topk_degenerate(eng):
    # all equal: the k largest indices; all NaN: dropped (count 0) or the k largest indices when kept
    x = np.full(5000, 3.5)
    v, i, _, _ = run_topk(eng, x, 10, 0)
    np.testing.assert_array_equal(i, np.arange(4999, 4989, -1))
    x = np.full(5000, np.nan)
    v, i, vpad, ipad = run_topk(eng, x, 10, 0)
    assert len(i) == 0 and len(ipad) == 10
    v, i, _, _ = run_topk(eng, x, 10, 1)
    np.testing.assert_array_equal(i, np.arange(4999, 4989, -1))
    # fewer rankable entries than k
    x = np.full(5000, np.nan)
    x[[7, 4000, 123]] = [1.0, 2.0, 1.0]
    v, i, vpad, ipad = run_topk(eng, x, 10, 0)
    np.testing.assert_array_equal(i, [4000, 123, 7])
    assert len(ipad) == 7


def test_nanmax_large(eng):
    _lib, H = eng
    rng = np.random.default_rng(0)
    for n in (4097, 300001):
        x = rng.standard_normal(n)
        x[rng.integers(0, n, n // 10)] = np.nan
        out = torch.empty(1, dtype=torch.float64, device="cuda")
        _lib.check(H.lib.gpimhip_nanmax(H.h, _lib.ptr(torch.from_numpy(x).cuda()), n, _lib.ptr(out)))
        assert out.item() == np.nanmax(x)
    x = np.full(10000, np.nan)
    out = torch.empty(1, dtype=torch.float64, device="cuda")
    _lib.check(H.lib.gpimhip_nanmax(H.h, _lib.ptr(torch.from_numpy(x).cuda()), len(x), _lib.ptr(out)))
    assert np.isnan(out.item())


def test_update_points_device_vs_reference(eng, golden_dir):
    """boptimizer.update_points (thinning loop on the GPU) reproduces the reference's outputs captured in
    tests/golden/host_logic.npz (cKDTree ball queries, boptim.py:326-376)."""
    from gpim_amd.boptim import boptimizer
    fx = dict(np.load(os.path.join(golden_dir, "host_logic.npz")))
    _lib, H = eng

    class Surrogate:
        _handle = H
    acq = 0.3 * fx["acq_mf"] + 1.7 * fx["acq_sf"]
    order = np.argsort(acq.ravel())[::-1][:35]
    vals = acq.ravel()[order].tolist()
    inds = np.stack(np.unravel_index(order, acq.shape), -1).tolist()

    def bare(**kw):
        bo = boptimizer.__new__(boptimizer)
        bo.verbose, bo.batch_update, bo.batch_size = 0, True, 35
        bo.dscale, bo.batch_dscale, bo.batch_out_max = None, None, kw["batch_out_max"]
        bo.gamma, bo.points_mem, bo.exit_strategy, bo.mask = 0.8, 10, 0, None
        bo.indices_all, bo.vals_all, bo.surrogate_model = [], [], Surrogate()
        return bo
    v, i = bare(batch_out_max=4).update_points(vals, inds, 1.5)
    assert_allclose(v, fx["bu_vals"], rtol=1e-15)
    assert i == fx["bu_inds"].tolist()
    v, i = bare(batch_out_max=3).update_points(vals, inds, 2.5)
    assert_allclose(v, fx["bul_vals"], rtol=1e-15)
    assert i == fx["bul_inds"].tolist()


def test_batch_update_with_random_padding_is_seeded(ensure_built, tmp_path):
    """batch_update=True with a batch so small that update_points pads it with np.random draws: two runs with
    the same seed give the same queries (reconstructor seeds numpy like pyro.set_rng_seed, gpr.py:102)."""
    import gpim_amd as gpim
    from problems import bo_test_problem
    runs = []
    for rep in range(2):
        trial_func, Z = bo_test_problem()
        bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), trial_func,
                             acquisition_function="cb", exploration_steps=3, batch_update=True, batch_size=6,
                             batch_out_max=5, batch_dscale=6.0, gp_iterations=50, verbose=0, seed=3,
                             filename=str(tmp_path / "bo"))
        bo.run()
        runs.append([tuple(i) for i in bo.indices_all])
    assert runs[0] == runs[1] and len(runs[0]) == 15


def test_bo_on_large_grid_device_resident(ensure_built, tmp_path):
    """EI exploration on a 256 x 256 grid (M = 65536: radix top-k, two-stage nanmax, lazy prediction maps):
    the ranking agrees with numpy on the materialised maps, and nothing but the ranked pairs is copied per
    step (the maps are materialised only when read)."""
    import gpim_amd as gpim
    from gpim_amd.boptim import _LazyMaps
    from scipy.stats import norm
    rng = np.random.default_rng(1)
    ii, jj = np.meshgrid(np.arange(256.), np.arange(256.), indexing="ij")
    truth = np.exp(-((ii - 60) ** 2 + (jj - 180) ** 2) / 800.0) + 0.6 * np.exp(-((ii - 190) ** 2 + (jj - 70) ** 2) / 1500.0)
    Z = np.full((256, 256), np.nan)
    seed = rng.integers(0, 256, size=(12, 2))
    Z[seed[:, 0], seed[:, 1]] = truth[seed[:, 0], seed[:, 1]]
    bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), lambda idx: truth[tuple(idx)],
                         acquisition_function="ei", exploration_steps=4, gp_iterations=100, verbose=0,
                         lengthscale=[[1., 1.], [60., 60.]], filename=str(tmp_path / "bo"))
    bo.single_step(0)
    assert isinstance(list.__getitem__(bo.gp_predictions, 0), _LazyMaps._Pending)      # still on the device
    mean, sd = bo.gp_predictions[0]                                                    # materialises
    assert mean.shape == (256, 256) and np.isfinite(mean).all() and (sd > 0).all()
    for e in range(1, 4):
        bo.single_step(e)
    assert len(bo.indices_all) == 4 and len(set(map(tuple, bo.indices_all))) == 4
    # the last step's choice is the EI arg-max of its own (materialised) posterior
    mean, sd = bo.gp_predictions[-1]
    obs = ~np.isnan(bo.target_func_vals[-2])
    imp = mean - mean[obs].max() - 0.01
    z = imp / sd
    ei = imp * norm.cdf(z) + sd * norm.pdf(z)
    top = np.unravel_index(np.argsort(ei.ravel())[::-1][:5], ei.shape)
    assert tuple(bo.indices_all[-1]) in set(zip(top[0].tolist(), top[1].tolist()))
