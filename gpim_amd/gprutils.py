"""
gprutils.py -- grid and data-layout helpers of the hot path (exported as ``gpim_amd.utils``).

Host-side mirror of the data-prep part of the reference's gpim/gprutils.py:23-210
(SURVEY 8(a) rows a1-a3): ``(c, *dims)`` coordinate grids <-> ``(N, c)`` row-major point
lists, NaN filtering, full / sparse index grids.  Plotting and corruption helpers of the
reference file (:213-938) are outside the hot path and not provided.
"""
import numpy as np
import torch


def _np_dtype(precision):
    return np.float32 if precision == "single" else np.float64


def prepare_training_data(X, y=None, vector_valued=False, **kwargs):
    """(c,*dims) grid + (*dims) observations -> torch tensors X:(N,c), y:(N,).

    Rows of X with any NaN coordinate are dropped and NaN entries of y are dropped; the
    two NaN patterns are expected to coincide (true for grids from ``get_sparse_grid``).
    Row-major order is preserved (reference gprutils.py:23-59).
    """
    dt = _np_dtype(kwargs.get("precision", "double"))
    pts = np.asarray(X).reshape(X.shape[0], -1).T
    pts = pts[~np.isnan(pts).any(axis=1)]
    Xt = torch.from_numpy(np.ascontiguousarray(pts, dtype=dt))
    if y is None:
        return Xt, y
    if vector_valued:
        yv = np.asarray(y).reshape(-1, y.shape[-1])
        yv = yv[~np.isnan(yv).any(axis=1)]
    else:
        yv = np.asarray(y).ravel()
        yv = yv[~np.isnan(yv)]
    return Xt, torch.from_numpy(np.ascontiguousarray(yv, dtype=dt))


def prepare_test_data(X, **kwargs):
    """(c,*dims) -> (M,c) torch tensor; NaN rows are kept (reference gprutils.py:62-85)."""
    dt = _np_dtype(kwargs.get("precision", "double"))
    pts = np.asarray(X).reshape(X.shape[0], -1).T
    return torch.from_numpy(np.ascontiguousarray(pts, dtype=dt))


def get_full_grid(R, extent=None, dense_x=1.):
    """Index coordinates of a 2D-4D array as an array of shape (ndim, *grid)
    (reference gprutils.py:108-172).  ``dense_x`` < 1 refines the grid; ``extent``
    ([[lo, hi], ...] per dimension) places it in physical units.  The reference only
    works with ``extent`` in 2D (its 3D/4D branches fail to unpack); here every
    dimensionality follows the 2D rule."""
    nd = np.ndim(R)
    if nd < 2 or nd > 4:
        raise NotImplementedError("Currently works only for 2D-4D sets")
    dense_x = np.float64(dense_x)
    if extent:
        axes = []
        for n, (lo, hi) in zip(np.shape(R), extent):
            axes.append(slice(lo, hi, dense_x / (n // (hi - lo))))
    else:
        axes = [slice(None, n, dense_x) for n in np.shape(R)]
    return np.array(np.mgrid[tuple(axes)])


def get_sparse_grid(R, extent=None):
    """Full grid with NaN coordinates at the missing observations of R
    (reference gprutils.py:175-210; 2D, and 3D with xy- or xyz-sparsity)."""
    if not np.isnan(R).any():
        raise NotImplementedError(
            "Missing values in sparse data must be represented as NaNs")
    nd = np.ndim(R)
    if nd not in (2, 3):
        raise NotImplementedError(
            "Currently supports only 2D and 3D sets with sparsity in xy and xyz dims")
    grid = get_full_grid(R, extent)
    missing = np.isnan(R)
    if nd == 3 and not missing[..., -1].any():
        # xy-sparsity: a NaN anywhere in a spectrum removes the whole (x, y) column
        missing = np.broadcast_to(missing.any(axis=-1, keepdims=True), R.shape)
    X = grid.copy()
    X[:, missing] = np.nan
    return X


def get_grid_indices(R, dense_x=1.):
    """(X_full, X_sparse) for a 2D/3D array (reference gprutils.py:88-105)."""
    if np.ndim(R) > 3:
        raise NotImplementedError("Currently supports only 2D and 3D arrays")
    return get_full_grid(R, dense_x=np.float64(dense_x)), get_sparse_grid(R)


def reflection_blocks(X, y, axes):
    """Symmetry reduction of an exact GP on a COMPLETE grid (gpim_amd extension; role of the reference's structured class
    gpim/gpreg/skgpr.py:399-448 for kernels that do not factorise over the axes -- csrc/engine.hip: kmat_refl_kernel).

    X (d, n_1, ..., n_d) grid coordinates, y (n_1, ..., n_d) observations, axes: the d coordinate vectors.  Every axis whose
    coordinates are symmetric about their centre is reflected; with r such axes the covariance of a stationary kernel that is
    even in each coordinate difference is block diagonal in the basis
        v_{s,p} = (|G| |Stab_p|)^-1/2 sum_g chi_s(g) e_{g p}      (p in the fundamental domain, s one of 2^r sign patterns)
    with blocks  K_s[p, q] = w_p w_q sum_g chi_s(g) k(p, g q),  w_p = |Stab_p|^-1/2.
    Returns a dict: mask (bit k = axis k reflected), twoc (first + last coordinate per axis), B = 2^r, Xq (the fundamental
    domain, (Nq, d): the first half of every reflected axis, including the mirror plane of an axis of odd length), ys (B, Nq):
    y in the adapted basis, wts (B, Nq) or None when no point lies on a mirror plane: w_p, and 0 where the point does not
    exist in the block (it lies on the mirror plane of an axis whose sign is -1), n_total = y.size.
    Raises ValueError if no axis is symmetric."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    d = X.shape[0]
    mask, twoc, dims = 0, [0.0] * 4, []
    for k, c in enumerate(axes):
        c = np.asarray(c, dtype=np.float64)
        if len(c) >= 2 and np.allclose(c + c[::-1], c[0] + c[-1], rtol=0, atol=1e-12 * max(1.0, abs(c[-1]), abs(c[0]))):
            mask |= 1 << k
            twoc[k] = float(c[0] + c[-1])
            dims.append(k)
    if not dims:
        raise ValueError("needs at least one grid axis with coordinates that are symmetric about their centre")
    B = 1 << len(dims)
    fund = tuple(slice(0, (y.shape[k] + 1) // 2) if k in dims else slice(None) for k in range(d))
    Xq = X[(slice(None),) + fund].reshape(d, -1).T.copy()
    fshape = y[fund].shape
    idx = np.indices(fshape)
    # on_plane[j]: the points of the domain on the mirror plane of the j-th reflected axis (odd length only)
    on_plane = [(idx[k] == y.shape[k] // 2) & (y.shape[k] % 2 == 1) for k in dims]
    nplanes = np.sum(on_plane, axis=0)
    w_pt = 2.0 ** (-0.5 * nplanes)                      # 1 / sqrt(|stabiliser|)
    ys, wts = np.empty((B, Xq.shape[0])), np.empty((B, Xq.shape[0]))
    for b in range(B):
        acc = np.zeros(fshape)
        for g in range(B):
            ax = tuple(dims[j] for j in range(len(dims)) if (g >> j) & 1)
            chi = -1.0 if bin(g & b).count("1") & 1 else 1.0
            acc += chi * (np.flip(y, axis=ax) if ax else y)[fund]
        present = np.ones(fshape, dtype=bool)
        for j in range(len(dims)):
            if (b >> j) & 1:
                present &= ~on_plane[j]                 # antisymmetric along an axis: nothing on its mirror plane
        wb = np.where(present, w_pt, 0.0)
        ys[b] = (acc * wb).reshape(-1) / np.sqrt(B)
        wts[b] = wb.reshape(-1)
    # for predictions on the training grid: the domain's points as flat grid indices, and every grid point's representative
    # in the domain (the posterior variance is invariant under the reflections)
    full = np.indices(y.shape)
    fund_flat = np.ravel_multi_index(tuple(full[k][fund] for k in range(d)), y.shape).reshape(-1)
    rep_idx = tuple(np.minimum(full[k], y.shape[k] - 1 - full[k]) if k in dims else full[k] for k in range(d))
    rep = np.ravel_multi_index(rep_idx, fshape).reshape(-1)
    return {"mask": mask, "twoc": twoc, "B": B, "Xq": Xq, "ys": ys, "wts": wts if nplanes.any() else None,
            "n_total": int(y.size), "dims": dims, "fund_flat": fund_flat, "rep": rep}

