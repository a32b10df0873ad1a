"""
Pins the oracle (oracle/gpim_oracle.py) to the reference's own known answers.

* test_bo_golden: the reference's test/test_boptim.py:42-58 re-run on the oracle, asserted
  against the reference's golden vectors test/test_data/test_{ei,poi,cb}.npy (copied to
  tests/golden/).  Also checks the query ORDER listed in SURVEY App. B.1.
* test_notebook_trace: the hyper-parameters printed after every 1000-iteration training in
  examples/notebooks/GP_based_exploration_exploitation.ipynb (four runs: plain EI, EI + mask,
  EI + dscale/memory, custom acquisition).  The CPU suite replays the first rows of each run;
  tests/tools/pin_oracle_full.py replays all 51 rows (result: tests/golden/oracle_full_replay.json).
"""
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from oracle import gpim_oracle as O
from problems import bo_test_problem, notebook_problem

ORDER = {
    "ei": [(24, 24), (0, 0), (24, 0), (14, 24), (0, 9), (0, 12), (24, 16), (0, 6), (6, 10), (2, 10),
           (6, 5), (7, 11), (5, 10), (5, 9), (5, 11), (16, 16), (10, 8), (10, 19), (18, 0), (19, 21)],
    "poi": [(24, 24), (0, 0), (24, 0), (14, 24), (0, 9), (0, 12), (24, 16), (0, 6), (6, 10), (4, 10),
            (14, 10), (5, 10), (5, 9), (5, 11), (5, 8), (17, 17), (10, 19), (6, 0), (18, 0), (8, 24)],
    "cb": [(24, 24), (0, 0), (24, 0), (14, 24), (0, 9), (24, 17), (10, 14), (0, 24), (24, 5), (16, 19),
           (0, 4), (0, 14), (24, 13), (24, 21), (12, 0), (10, 11), (0, 16), (0, 6), (24, 2), (11, 17)],
}


@pytest.mark.parametrize("acqf", ["ei", "poi", "cb"])
def test_bo_golden(acqf, golden_dir):
    trial_func, Z_sparse = bo_test_problem()
    bo = O.boptimizer(O.get_sparse_grid(Z_sparse), Z_sparse, O.get_full_grid(Z_sparse), trial_func,
                      acquisition_function=acqf, exploration_steps=20, use_gpu=False, verbose=0)
    bo.run()
    expected = np.load(os.path.join(golden_dir, "test_%s.npy" % acqf))
    assert_allclose(bo.target_func_vals[-1], expected)         # the reference's own assertion
    assert [tuple(i) for i in bo.indices_all] == ORDER[acqf]


def run_notebook(which, nsteps, factory):
    """Replays one of the notebook's BO runs; returns the printed (rounded) rows."""
    trial_func, Z_sparse = notebook_problem(5)
    kw, af = {}, "ei"
    if which == "ei_mask":
        mask = np.ones((25, 25)) * np.nan
        mask[2:23, 2:23] = 1
        kw["mask"] = mask
    elif which == "ei_dscale":
        kw.update(dscale=4, memory=10)
    elif which == "custom":
        def af(gpmodel, X_full, X_sparse):
            mean, sd = gpmodel.predict(X_full, verbose=0)
            return 1 * mean + 5 * sd, (mean, sd)
    bo, getter = factory(Z_sparse, trial_func, af, nsteps, kw)
    rows = []
    train = bo.surrogate_model.train

    def recording_train(**k):
        train(**k)
        rows.append(getter(bo))
    bo.surrogate_model.train = recording_train
    bo.run()
    return np.array(rows), bo


def check_rows(rows, expected):
    expected = np.array(expected[:len(rows)])
    # printed with 4 decimals (amp, lengthscales) / 7 decimals (noise)
    tol = np.array([1.01e-4, 1.01e-4, 1.01e-4, 1.01e-7])
    exact = (np.abs(rows - expected) <= tol).all(axis=1)
    # training stops on flat directions now and then: a late digit may differ between torch
    # builds without changing the chosen points (all later rows match again)
    assert (~exact).sum() <= max(1, len(rows) // 8), (rows, expected)
    assert_allclose(rows[:, :3], expected[:, :3], rtol=1e-2, atol=2e-4)
    assert exact[0] and exact[1]


def oracle_factory(Z_sparse, trial_func, af, nsteps, kw):
    bo = O.boptimizer(O.get_sparse_grid(Z_sparse), Z_sparse, O.get_full_grid(Z_sparse), trial_func,
                      acquisition_function=af, exploration_steps=nsteps, use_gpu=False, verbose=0, **kw)

    def getter(b):
        k = b.surrogate_model.kernel
        return [np.around(k.variance.item(), 4), *np.around(k.lengthscale.tolist(), 4),
                np.around(k.noise.item(), 7)]
    return bo, getter


@pytest.mark.parametrize("which,nsteps", [("ei", 12), ("ei_mask", 6), ("ei_dscale", 6), ("custom", 6)])
def test_notebook_trace(which, nsteps, golden_dir):
    trace = json.load(open(os.path.join(golden_dir, "notebook_trace.json")))["runs"][which]
    rows, bo = run_notebook(which, nsteps, oracle_factory)
    assert len(rows) == nsteps + 1
    check_rows(rows, trace)
    if which == "ei":    # first eight queried points
        assert [tuple(i) for i in bo.indices_all[:8]] == [(14, 3), (14, 12), (12, 9), (11, 9), (10, 8),
                                                         (9, 7), (8, 8), (9, 8)]


def test_init_draw_constants():
    """seed 0, d=2, default bounds on a 25x25 grid."""
    import torch
    torch.manual_seed(0)
    kp = O.KernelParams("RBF", 2, [[0., 0.], [12.5, 12.5]])
    assert kp.variance.item() == pytest.approx(9.70053301276535, rel=1e-14)
    assert kp.lengthscale.tolist() == pytest.approx([8.84774830499735, 5.742286789093136], rel=1e-14)
    assert kp.noise.item() == 1.0


def test_oracle_sparse_gp_properties():
    """No reference known answer exists for the sparse (VFE) model, so the oracle's restatement is
    at least checked against identities of the model itself: (i) with every observation used as an
    inducing input the bound is tight -- VFE loss == exact negative log marginal likelihood and the
    posteriors coincide; (ii) fewer inducing inputs can only raise the loss (it is an upper bound
    of the exact NLL); (iii) autograd gradient == central finite differences."""
    import torch
    torch.manual_seed(3)
    rng = np.random.default_rng(0)
    X = torch.from_numpy(np.unique(rng.integers(0, 12, size=(80, 2)), axis=0).astype(np.float64))[:40]
    y = torch.from_numpy(np.sin(X.numpy().sum(1) / 3.0) + 0.05 * rng.standard_normal(len(X)))
    kp = O.KernelParams("RBF", 2, [[1., 1.], [6., 6.]])
    with torch.no_grad():
        kp.u_noise.fill_(-3.0)
    exact = O.ExactGP(X, y, kp, 0.0)
    full = O.SparseGP(X, y, kp, X.clone(), 1e-10)
    assert_allclose(full.loss().item(), exact.loss().item(), rtol=1e-6)
    Xs = torch.from_numpy(rng.uniform(0, 12, size=(50, 2)))
    me, ve = exact.predict(Xs)
    ms, vs = full.predict(Xs)
    assert_allclose(ms.numpy(), me.numpy(), atol=1e-5)
    assert_allclose(vs.numpy(), ve.numpy(), atol=1e-5)
    few = O.SparseGP(X, y, kp, X[::5].clone(), 1e-10)
    assert few.loss().item() >= exact.loss().item() - 1e-9
    # gradient check on the inducing inputs and the noise
    loss0, g = few.loss_and_grad()
    P = 4
    eps = 1e-6
    for idx in (0, 3, 7):
        with torch.no_grad():
            few.Xu.view(-1)[idx] += eps
            lp = few.loss().item()
            few.Xu.view(-1)[idx] -= 2 * eps
            lm = few.loss().item()
            few.Xu.view(-1)[idx] += eps
        assert_allclose((lp - lm) / (2 * eps), g[P + idx].item(), rtol=1e-5, atol=1e-6)
    with torch.no_grad():
        kp.u_noise += eps
        lp = few.loss().item()
        kp.u_noise -= 2 * eps
        lm = few.loss().item()
        kp.u_noise += eps
    assert_allclose((lp - lm) / (2 * eps), g[3].item(), rtol=1e-5)
