"""
Symmetry-reduced exact GP on complete grids (reconstructor(structured=True) with Matern52 / RationalQuadratic; role of the
reference's structured class gpim/gpreg/skgpr.py:399-448, exact instead of interpolated) against the DENSE oracle: the
reflection-adapted basis is an orthogonal change of basis, so losses, hyper-parameter histories and posteriors must
agree with the dense model to rounding.
"""
import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

from oracle import gpim_oracle as O


@pytest.fixture(scope="module")
def gpim(ensure_built):
    import gpim_amd
    return gpim_amd


def _image(shape, seed):
    rng = np.random.default_rng(seed)
    grids = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
    R = np.ones(shape)
    for k, g in enumerate(grids):
        R = R * np.cos(g / (2.0 + k) + 0.3 * k)
    return R + 0.05 * rng.standard_normal(shape)


@pytest.mark.parametrize("shape,kernel,kw", [
    ((12, 10), "Matern52", dict(lengthscale=[[1., 1.], [6., 6.]])),          # both axes reflected: 4 blocks of 30
    ((12, 9), "Matern52", dict(lengthscale=[[1., 1.], [6., 6.]])),           # an odd axis: its mirror plane is in the domain
    ((9, 7), "Matern52", dict(lengthscale=[[1., 1.], [5., 5.]])),            # both odd: points on one and on two planes
    ((7, 6, 5), "RationalQuadratic", dict(lengthscale=[[1., 1., 1.], [4., 4., 4.]])),
    ((18, 16), "RationalQuadratic", dict(lengthscale=[[1., 1.], [8., 8.]])),  # 4 blocks of 72
    ((6, 4, 8), "Matern52", dict(lengthscale=[[1., 1., 1.], [4., 4., 4.]])),  # 3-D: 8 blocks of 24
    ((16, 14), "Matern52", dict(lengthscale=[1., 6.], isotropic=True)),
])
def test_symmetry_reduced_vs_dense_oracle(gpim, shape, kernel, kw):
    R = _image(shape, seed=len(shape) + shape[0])
    X = gpim.utils.get_full_grid(R)
    T = 12
    args = dict(kernel=kernel, learning_rate=0.1, iterations=T, verbose=0, **kw)
    rec = gpim.reconstructor(X, R, X, structured=True, **args)
    assert rec.do_symm and not rec.do_structured
    mean, sd, hyper = rec.run()
    mo, so, ho = O.reconstructor(X, R, X, **args).run()
    assert_allclose(rec.loss_all, O_losses(X, R, args), rtol=1e-10)
    for k in ("lengthscale", "noise", "variance"):
        assert_allclose(np.asarray(hyper[k], dtype=float), np.asarray(ho[k], dtype=float), rtol=1e-8)
    assert np.abs(mean - mo).max() < 1e-8 and np.abs(sd - so).max() < 1e-8
    # the dense HIP path gives the same answers
    md, sdd, hd = gpim.reconstructor(X, R, X, **args).run()
    assert np.abs(mean - md).max() < 1e-8 and np.abs(sd - sdd).max() < 1e-8
    # prediction at points off the grid (no product structure needed)
    Xoff = np.stack([g.ravel()[:50] + 0.37 for g in X]).reshape((len(shape), 50))
    m2, s2 = rec.predict(Xoff)
    orc = O.reconstructor(X, R, X, **args)
    orc.train()
    m2o, s2o = orc.predict(Xoff)
    assert np.abs(m2 - m2o).max() < 1e-8 and np.abs(s2 - s2o).max() < 1e-8


def test_symmetry_reduced_64x64_vs_dense_oracle(gpim):
    """A 64 x 64 image (N = 4096: four blocks of 1024, the blocked general path with eight block columns)."""
    R = _image((64, 64), seed=5)
    X = gpim.utils.get_full_grid(R)
    args = dict(kernel="Matern52", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=3, verbose=0)
    rec = gpim.reconstructor(X, R, X, structured=True, **args)
    mean, sd, hyper = rec.run()
    import torch
    torch.set_num_threads(min(32, torch.get_num_threads() * 32))
    orc = O.reconstructor(X, R, X, **args)
    mo, so, ho = orc.run()
    torch.set_num_threads(1)
    assert_allclose(rec.loss_all, orc.loss_all, rtol=1e-10)
    for k in ("lengthscale", "noise", "variance"):
        assert_allclose(np.asarray(hyper[k], dtype=float), np.asarray(ho[k], dtype=float), rtol=1e-8)
    assert np.sqrt(np.mean((mean - mo) ** 2)) < 1e-8 and np.sqrt(np.mean((sd - so) ** 2)) < 1e-8


def O_losses(X, R, args):
    orc = O.reconstructor(X, R, X, **args)
    orc.train()
    return np.asarray(orc.loss_all, dtype=float)


def test_skreconstructor_matern(gpim):
    """The structured class of the reference takes Matern52 (gpim/gpreg/skgpr.py:399-448, gpytorch_kernels.py:65)."""
    R = _image((14, 10), 2)
    X = gpim.utils.get_full_grid(R)
    kw = dict(kernel="Matern52", lengthscale=[[1., 1.], [6., 6.]], learning_rate=0.1, iterations=8, verbose=0)
    mean, sd, hyper = gpim.skreconstructor(X, R, X, **kw).run()
    mo, so, ho = O.reconstructor(X, R, X, **kw).run()
    assert_allclose(np.asarray(hyper["lengthscale"], dtype=float), np.asarray(ho["lengthscale"], dtype=float), rtol=1e-8)
    assert np.abs(mean - mo).max() < 1e-8 and np.abs(sd - so).max() < 1e-8


def test_sharded_blocks_world_1_equal_the_batched_model(gpim):
    """gpim_amd.dist_symm (the reflection blocks dealt to the ranks of a job) at world size 1: the histories of
    reconstructor(structured=True), the dense oracle's posterior.  (World size 2: tests/tools/dist2_worker.py.)"""
    from gpim_amd.dist_symm import symm_gp_fit, symm_gp_posterior
    R = _image((12, 9, 4), 3)
    X = gpim.utils.get_full_grid(R)
    kw = dict(kernel="Matern52", lengthscale=[[1., 1., 1.], [6., 6., 6.]])
    T = 8
    hyper, u = symm_gp_fit(X, R, learning_rate=0.1, iterations=T, **kw)
    rec = gpim.reconstructor(X, R, X, structured=True, learning_rate=0.1, iterations=T, verbose=0, **kw)
    rec.train()
    assert_allclose(hyper["loss"], rec.loss_all, rtol=1e-12)
    assert_allclose(hyper["lengthscale"], np.asarray(rec.hyperparams["lengthscale"]), rtol=1e-12)
    orc = O.reconstructor(X, R, X, learning_rate=0.1, iterations=T, verbose=0, **kw)
    mo, so, _ = orc.run()
    pts = X.reshape(3, -1).T
    mean, sd = symm_gp_posterior(X, R, pts, u, **kw)
    assert np.abs(mean - mo.ravel()).max() < 1e-8 and np.abs(sd - so.ravel()).max() < 1e-8
    # Xtest=None: the training grid, the variance on the fundamental domain and mirrored (12 even, 9 odd, 4 even)
    mean_g, sd_g = symm_gp_posterior(X, R, None, u, **kw)
    assert np.abs(mean_g - mo.ravel()).max() < 1e-8 and np.abs(sd_g - so.ravel()).max() < 1e-8
    # one shard for fit and posterior (shared workspace): the same numbers
    from gpim_amd.dist_symm import symm_shard
    sh = symm_shard(X, R, **kw)
    hyper2, u2 = symm_gp_fit(X, R, learning_rate=0.1, iterations=T, shard=sh, **kw)
    assert np.array_equal(hyper2["loss"], hyper["loss"])
    mean_s, sd_s = symm_gp_posterior(X, R, None, u2, shard=sh, **kw)
    assert np.array_equal(mean_s, mean_g) and np.array_equal(sd_s, sd_g)


def test_symmetry_reduced_needs_a_symmetric_axis(gpim):
    R = _image((9, 7), 1)
    X = gpim.utils.get_full_grid(R).astype(np.float64)
    X[0] = X[0] ** 1.5                                      # neither axis is symmetric about its centre any more
    X[1] = X[1] ** 1.5
    with pytest.raises(NotImplementedError):
        gpim.reconstructor(X, R, X, structured=True, kernel="Matern52", verbose=0)


def test_rank_without_a_block_runs_the_replicated_step(gpim):
    """World size > number of reflection blocks (a 2-D image has four, a node eight GPUs): a rank that owns no block never
    builds a matrix workspace, yet runs the replicated chain rule + Adam step of every iteration on the all-reduced sums
    (gpimhip_dist_finalize_dev used to refuse a handle without workspace: that rank raised while the others hung in the
    next all-reduce).  Single process: the shard of rank 5 of 8, whose sums are zero -- the step must run and keep the
    parameters finite; the dead-factor guard of DistributedCholesky rides along."""
    from gpim_amd.dist_symm import _Shard, symm_gp_fit, symm_gp_posterior
    from gpim_amd.kernels import KernelSpec
    R = _image((12, 10), 5)
    X = gpim.utils.get_full_grid(R)
    kw = dict(kernel="Matern52", lengthscale=[[1., 1.], [6., 6.]])
    spec = KernelSpec("Matern52", 2, kw["lengthscale"], jitter=1e-5)
    sh = _Shard(X, R, spec, 5, 8)
    assert sh.B == 4 and sh.mine == []
    hyper, u = symm_gp_fit(X, R, learning_rate=0.1, iterations=3, shard=sh, **kw)
    assert np.isfinite(hyper["loss"]).all() and np.isfinite(np.asarray(hyper["lengthscale"])).all() and torch.isfinite(u).all()
    mean, sd = symm_gp_posterior(X, R, None, u, shard=sh, **kw)          # this rank's share of the sums: zeros
    assert np.all(mean == 0.0) and np.isfinite(sd).all()
    from gpim_amd.dist_chol import DistributedCholesky
    A = torch.eye(700, dtype=torch.float64, device="cuda") * 3.0
    ch = DistributedCholesky(700).set_from_function(lambda c0, c1: A[:, c0:c1])
    with pytest.raises(RuntimeError):
        ch.solve(torch.ones(700, dtype=torch.float64, device="cuda"))      # no factor yet
    ch.factor()
    Xl = ch.inverse()
    ch.kinv(Xl, out=ch.local)                                             # K^-1 over the dead factor ...
    with pytest.raises(RuntimeError):
        ch.logdet()                                                       # ... which is gone
