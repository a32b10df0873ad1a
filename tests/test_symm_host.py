"""Host logic of the symmetry-reduced exact GP (gpim_amd.gprutils.reflection_blocks; the device side is
csrc/engine.hip: kmat_refl_kernel, tested against the dense oracle in tests/test_gpu_symm.py): with the blocks
K_s[p, q] = w_p w_q sum_g chi_s(g) k(p, g q) and the projected observations y_s, the dense model's log-determinant and
quadratic form -- hence its marginal likelihood -- must come out as the sums over the blocks, for even and odd axis lengths.
Pure numpy; no GPU."""
import numpy as np
import pytest

from gpim_amd import gprutils


def matern52(A, Bz, ls, var):
    D = (A[:, None, :] - Bz[None, :, :]) / ls
    r = np.sqrt((D ** 2).sum(-1) + 1e-12)
    return var * (1 + np.sqrt(5) * r + 5.0 / 3.0 * r ** 2) * np.exp(-np.sqrt(5) * r)


@pytest.mark.parametrize("shape", [(6, 4), (6, 5), (5, 7), (4, 3, 5), (5, 1)])
def test_reflection_blocks_reproduce_the_dense_model(shape):
    rng = np.random.default_rng(sum(shape))
    y = rng.standard_normal(shape)
    X = np.stack(np.meshgrid(*[np.arange(n, dtype=float) * (1.0 + 0.5 * k) for k, n in enumerate(shape)], indexing="ij"))
    axes = [np.arange(n, dtype=float) * (1.0 + 0.5 * k) for k, n in enumerate(shape)]
    d = len(shape)
    ls, var, noise = np.array([1.7, 2.3, 1.1][:d]), 1.3, 0.05
    pts = X.reshape(d, -1).T
    K = matern52(pts, pts, ls, var) + noise * np.eye(len(pts))
    sign, logdet = np.linalg.slogdet(K)
    quad = y.ravel() @ np.linalg.solve(K, y.ravel())
    if shape == (5, 1):
        axes[1] = np.array([0.0])                        # (a degenerate axis is not reflected)
    S = gprutils.reflection_blocks(X, y, axes)
    B, Xq = S["B"], S["Xq"]
    assert B == 2 ** len(S["dims"]) and S["n_total"] == y.size
    # sum_s |y_s|^2 = |y|^2: the basis is orthonormal
    assert np.isclose((S["ys"] ** 2).sum(), (y ** 2).sum(), rtol=1e-13)
    ld_sum, quad_sum, n_present = 0.0, 0.0, 0
    for b in range(B):
        w = S["wts"][b] if S["wts"] is not None else np.ones(len(Xq))
        Ks = np.zeros((len(Xq), len(Xq)))
        for g in range(B):
            Z = Xq.copy()
            chi = 1.0
            for j, k in enumerate(S["dims"]):
                if (g >> j) & 1:
                    Z[:, k] = S["twoc"][k] - Z[:, k]
                    if (b >> j) & 1:
                        chi = -chi
            Ks += chi * matern52(Xq, Z, ls, var)
        Ks *= w[:, None] * w[None, :]
        Ks[np.diag_indices_from(Ks)] += noise
        absent = w == 0
        Ks[absent, :] = 0.0; Ks[:, absent] = 0.0
        Ks[absent, absent] = 1.0                         # identity rows where the point does not exist in the block
        assert np.all(S["ys"][b][absent] == 0.0)
        n_present += int((~absent).sum())
        ld_sum += np.linalg.slogdet(Ks)[1]
        quad_sum += S["ys"][b] @ np.linalg.solve(Ks, S["ys"][b])
    assert n_present == y.size
    assert np.isclose(ld_sum, logdet, rtol=1e-11, atol=1e-10)
    assert np.isclose(quad_sum, quad, rtol=1e-10)


def test_reflection_blocks_reject_grids_without_a_symmetric_axis():
    y = np.zeros((4, 3))
    axes = [np.array([0.0, 1.0, 3.0, 7.0]), np.array([0.0, 1.0, 4.0])]
    X = np.stack(np.meshgrid(*axes, indexing="ij"))
    with pytest.raises(ValueError):
        gprutils.reflection_blocks(X, y, axes)
