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
