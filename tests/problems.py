"""Seeded problem generators shared by the tests, bench.py and the smoke check."""
import numpy as np


def bo_test_problem():
    """test/test_boptim.py:17-39 of the reference: one Gaussian on 25x25, 5 random seeds."""
    def trial_func(idx, **kwargs):
        x0, y0, fwhm = kwargs.get("x0", 5), kwargs.get("y0", 10), kwargs.get("fwhm", 4.5)
        return np.exp(-4 * np.log(2) * ((idx[0] - x0) ** 2 + (idx[1] - y0) ** 2) / fwhm ** 2)
    np.random.seed(0)
    x = np.arange(0, 25, 1.)
    y = x[:, np.newaxis]
    Z = trial_func([y, x])
    idx = np.random.randint(0, Z.shape[0], size=(2, 5))
    Z_sparse = np.ones_like(Z) * np.nan
    Z_sparse[idx[0], idx[1]] = Z[idx[0], idx[1]]
    return trial_func, Z_sparse


def notebook_problem(n_seed=5):
    """GP_based_exploration_exploitation.ipynb (n_seed=5) / README.md:71-106 (n_seed=4)."""
    def trial_func(idx):
        def func(x0, y0, a, b, fwhm):
            return np.exp(-4 * np.log(2) * (a * (idx[0] - x0) ** 2 + b * (idx[1] - y0) ** 2) / fwhm ** 2)
        return func(5, 10, 1, 1, 4.5) + func(10, 8, 0.75, 1.5, 7) + func(18, 18, 1, 1.5, 10)
    np.random.seed(42)
    Z_sparse = np.ones((25, 25)) * np.nan
    for i in np.random.randint(0, 25, size=(n_seed, 2)):
        Z_sparse[tuple(i)] = trial_func(i)
    return trial_func, Z_sparse


def gpr_dummy_data(seed=0):
    """test/test_gpreg.py:9-21 of the reference: 20x20 Gaussian with up to 200 NaNs."""
    rng = np.random.RandomState(seed)
    xx, yy = np.meshgrid(np.arange(0, 100, 5), np.arange(0, 100, 5))
    Z = np.exp(-((xx - 25) ** 2 + (yy - 50) ** 2) / 300)
    for _ in range(200):
        Z[rng.randint(Z.shape[0]), rng.randint(Z.shape[1])] = np.nan
    return Z


def spiral_image(size=128, keep=0.257, seed=0):
    """Synthetic twin of config C1: three Gaussians + noise on size x size,
    observed along an Archimedean spiral covering ~`keep` of the pixels."""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    def g(x0, y0, s):
        return np.exp(-((ii - x0) ** 2 + (jj - y0) ** 2) / (2 * s ** 2))
    img = g(0.3 * size, 0.35 * size, 0.12 * size) + 0.7 * g(0.7 * size, 0.6 * size, 0.18 * size) \
        + 0.5 * g(0.45 * size, 0.8 * size, 0.08 * size) + 0.01 * rng.standard_normal((size, size))
    img = (img - img.min()) / np.ptp(img)
    c = (size - 1) / 2.0
    r = np.hypot(ii - c, jj - c)
    th = np.arctan2(jj - c, ii - c)
    pitch = 1.0 / keep                       # spiral arm spacing in pixels (arm is ~1 px wide)
    arm = np.mod(r - pitch * th / (2 * np.pi), pitch)
    mask = (arm < 1.0) & (r <= c * 1.42)
    R = np.where(mask, img, np.nan)
    return R, img


def spiral_pfm_image():
    """Config C1 as the reference defines it (SURVEY 8(d) "Config 1"): the 128x128 PFM spiral scan
    ``expdata/spiral_s_00010_2019.npy`` (a data file of the reference, committed as the fixture
    tests/golden/spiral_s_00010_2019.npy), normalised (x - min) / ptp, the constant background set to NaN
    as examples/notebooks/GP_2D3D_images.ipynb does: N = 4212 observations, M = 16384."""
    import os
    img = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "spiral_s_00010_2019.npy"))
    R = (img - np.amin(img)) / np.ptp(img)
    R[R == R[1, 1]] = np.nan
    return R


def lattice_image(size=256, frac=0.25, seed=1):
    """Synthetic twin of config C2: twisted-bilayer hexagonal lattice image,
    `frac` of the pixels observed uniformly at random."""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    def layer(theta):
        out = np.zeros_like(ii)
        q = 2 * np.pi / 8.0
        for k in range(3):
            a = theta + k * 2 * np.pi / 3
            out += np.cos(q * (np.cos(a) * ii + np.sin(a) * jj))
        return out / 3.0
    img = layer(0.0) + layer(np.deg2rad(5.0)) + 0.02 * rng.standard_normal((size, size))
    img = (img - img.min()) / np.ptp(img)
    obs = np.random.default_rng(seed + 1).random((size, size)) < frac
    n_target = int(round(frac * size * size))
    # make the observation count exact (N = frac * size^2) for reproducible sizing
    flat = np.flatnonzero(obs)
    if len(flat) > n_target:
        drop = np.random.default_rng(seed + 2).choice(flat, len(flat) - n_target, replace=False)
        obs.flat[drop] = False
    elif len(flat) < n_target:
        free = np.flatnonzero(~obs)
        add = np.random.default_rng(seed + 2).choice(free, n_target - len(flat), replace=False)
        obs.flat[add] = True
    R = np.where(obs, img, np.nan)
    return R, img


def hyperspectral_cube(size=64, nspec=64, keep=0.30, seed=3):
    """Synthetic twin of config C3: A(x,y) * Lorentzian(l; l0(x,y), w) + noise; `keep` of the
    (x,y) columns observed (the same mask for every spectral slice)."""
    rng = np.random.default_rng(seed)
    ii, jj = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    A = 1.0 + 0.5 * np.sin(ii / 9.0) * np.cos(jj / 7.0)
    l0 = nspec / 2 + 0.15 * nspec * np.sin((ii + jj) / 15.0)
    ll = np.arange(nspec)[None, None, :]
    w = nspec / 10.0
    cube = A[..., None] * (w ** 2 / ((ll - l0[..., None]) ** 2 + w ** 2)) + 0.01 * rng.standard_normal((size, size, nspec))
    mask = rng.random((size, size)) < keep
    R = np.where(mask[..., None], cube, np.nan)
    return R, cube


def ckpfm_cube(nx=10, ny=10, nv=64, ns=5, seed=5):
    """Synthetic twin of config C5: a 10 x 10 x Nv x Ns cKPFM response -- a smooth,
    nearly separable 4-D function (hysteresis-like tanh in the voltage axis whose offset drifts over
    the (x, y) grid and with the read step) + noise; fully observed, like the reference's
    Nd_mat_amp * cos(Nd_mat_phase) cube (examples/notebooks/GP_TD_cKPFM.ipynb:332-339)."""
    rng = np.random.default_rng(seed)
    i, j, v, s = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nv), np.arange(ns), indexing="ij")
    v0 = nv / 2 + 0.12 * nv * np.sin(i / 3.0) * np.cos(j / 4.0) + 1.5 * (s - ns / 2)
    cube = (1.0 + 0.2 * np.cos((i + j) / 5.0)) * np.tanh((v - v0) / (0.12 * nv)) * (1.0 - 0.08 * s)
    return cube + 0.02 * rng.standard_normal(cube.shape)


class oracle_threads:
    """Lets the CPU oracle use the host's cores for one large comparison (the suite otherwise runs
    torch single-threaded, which is fastest for the many tiny-N oracle loops)."""

    def __init__(self, n=32):
        self.n = n

    def __enter__(self):
        import os
        import torch
        self.prev = torch.get_num_threads()
        torch.set_num_threads(max(1, min(self.n, os.cpu_count() or 1)))

    def __exit__(self, *a):
        import torch
        torch.set_num_threads(self.prev)
