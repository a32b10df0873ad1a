"""
Pins the oracle's Matern52 / RationalQuadratic exact GPs and its sparse VFE model -- for which the
reference holds no known answers -- to an independent 50-digit evaluation of the published formulas
(tests/golden/gp_highprec.npz, written by tests/tools/make_highprec_fixtures.py with mpmath: dense linear
algebra only, finite-difference gradients, no code or algebra shared with the oracle).
"""
import os

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

from oracle import gpim_oracle as O


def load_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "gp_highprec.npz"))
    for ci in range(int(z["n_cases"])):
        t = "c%d_" % ci
        yield {k[len(t):]: z[k] for k in z.files if k.startswith(t)}


def oracle_kernel(c):
    kind, d = str(c["kind"]), c["X"].shape[1]
    kp = O.KernelParams(kind, d, [c["ls"][0].tolist(), c["ls"][1].tolist()])
    u = torch.from_numpy(c["u"])
    with torch.no_grad():
        kp.u_var.copy_(u[0])
        kp.u_ls.copy_(u[1:1 + d])
        kp.u_noise.copy_(u[1 + d])
        if kp.u_alpha is not None:
            kp.u_alpha.copy_(u[2 + d])
    return kp


def test_fixture_present_and_shaped(golden_dir):
    cases = list(load_cases(golden_dir))
    assert len(cases) >= 4 and {str(c["kind"]) for c in cases} == {"RBF", "Matern52", "RationalQuadratic"}


def test_exact_gp_vs_high_precision(golden_dir):
    for c in load_cases(golden_dir):
        kp = oracle_kernel(c)
        X, y = torch.from_numpy(c["X"]), torch.from_numpy(c["y"])
        gp = O.ExactGP(X, y, kp, float(c["jitter"]))
        loss, g = gp.loss_and_grad()
        assert_allclose(loss.item(), float(c["loss"]), rtol=1e-13)
        assert_allclose(g.numpy(), c["grad"], rtol=1e-10, atol=1e-12)
        mean, var = gp.predict(torch.from_numpy(c["Xs"]))
        assert_allclose(mean.numpy(), c["mean"], rtol=0, atol=1e-12)
        assert_allclose(var.numpy(), c["var"], rtol=0, atol=1e-12)


def test_sparse_vfe_vs_high_precision(golden_dir):
    for c in load_cases(golden_dir):
        kp = oracle_kernel(c)
        X, y = torch.from_numpy(c["X"]), torch.from_numpy(c["y"])
        gp = O.SparseGP(X, y, kp, torch.from_numpy(c["Xu"]), float(c["jitter"]))
        loss, g = gp.loss_and_grad()
        assert_allclose(loss.item(), float(c["vfe_loss"]), rtol=1e-12)
        assert_allclose(g.numpy(), c["vfe_grad"], rtol=1e-8, atol=1e-10)
        mean, var = gp.predict(torch.from_numpy(c["Xs"]))
        assert_allclose(mean.numpy(), c["vfe_mean"], rtol=0, atol=1e-10)
        assert_allclose(var.numpy(), c["vfe_var"], rtol=0, atol=1e-10)


def load_cases2(golden_dir):
    """tests/golden/gp_highprec2.npz (tests/tools/make_highprec_fixtures2.py): a d = 4 Matern52 GP and two GPs with
    an ISOTROPIC lengthscale (d = 3), exact models only."""
    z = np.load(os.path.join(golden_dir, "gp_highprec2.npz"))
    for ci in range(int(z["n_cases"])):
        t = "c%d_" % ci
        yield {k[len(t):]: z[k] for k in z.files if k.startswith(t)}


def test_isotropic_and_4d_vs_high_precision(golden_dir):
    cases = list(load_cases2(golden_dir))
    assert any(bool(c["iso"]) for c in cases) and any(c["X"].shape[1] == 4 for c in cases)
    for c in cases:
        kind, d, iso = str(c["kind"]), c["X"].shape[1], bool(c["iso"])
        ls = [float(c["ls"][0]), float(c["ls"][1])] if iso else [c["ls"][0].tolist(), c["ls"][1].tolist()]
        kp = O.KernelParams(kind, d, ls)
        u = torch.from_numpy(c["u"])
        nl = 1 if iso else d
        with torch.no_grad():
            kp.u_var.copy_(u[0])
            kp.u_ls.copy_(u[1:1 + nl].reshape(kp.u_ls.shape))
            kp.u_noise.copy_(u[1 + nl])
            if kp.u_alpha is not None:
                kp.u_alpha.copy_(u[2 + nl])
        gp = O.ExactGP(torch.from_numpy(c["X"]), torch.from_numpy(c["y"]), kp, float(c["jitter"]))
        loss, g = gp.loss_and_grad()
        assert_allclose(loss.item(), float(c["loss"]), rtol=1e-13)
        assert_allclose(g.numpy(), c["grad"], rtol=1e-10, atol=1e-12)
        mean, var = gp.predict(torch.from_numpy(c["Xs"]))
        assert_allclose(mean.numpy(), c["mean"], rtol=0, atol=1e-12)
        assert_allclose(var.numpy(), c["var"], rtol=0, atol=1e-12)
