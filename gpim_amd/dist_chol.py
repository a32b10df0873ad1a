"""
dist_chol.py -- ONE exact GP across the GPUs of a node: block-column-cyclic Cholesky of the covariance
over ``torch.distributed`` (RCCL over xGMI), SURVEY 8(f) rank 1.

The reference fits a whole cube as one d-dimensional GP (gpim/gpreg/gpr.py:30-43,115-126); its dense
covariance stops fitting one device beyond N ~ 10^5 (N = 65536, the complete 256 x 256 image of config
C2, is 32 GiB).  Here the N x N matrix is dealt to the P ranks by 512-column panels -- panel p belongs to
rank p mod P, the 1 x P case of a 2-D block-cyclic layout: every rank owns whole columns, so a panel is
factored by its owner alone and ONE broadcast per panel (np x 512 doubles, 268 MB at N = 65536) is the
only data-path collective.  Round p:

    owner(p) : factor panel p in place              (C ABI: gpimhip_dist_panel_factor -- the potf2 /
                                                      panel-solve / in-panel-update chain of csrc/potf2.hip)
    all      : broadcast of the factored panel       (dist.broadcast, src = owner(p))
    each     : trailing update of every owned panel right of p with the broadcast copy
                                                     (gpimhip_dist_trailing_update -- the fp64 MFMA tile engine)

The owner of panel p+1 updates that panel FIRST, so its factorisation and broadcast can start while the
other ranks are still updating (look-ahead by ordering; RCCL runs the broadcast on its own stream).
Per-rank memory: N^2 / P doubles + two panel buffers.  Communication per rank: N^2 / 2 doubles received in
total, about the time of the 1/P share of the N^3 / 3 flop at N = 65536, P = 8.

What the distributed model offers: the factor, log det K, the negative log marginal likelihood at given
hyper-parameters and the posterior mean and standard deviation (two distributed triangular solves for
alpha, O(N^2); K*^T alpha sharded over test points; for the variance the factor is streamed through every
rank once more while each rank forward-substitutes its own test columns).  Training gradients (the
distributed K^-1) are not built: hyper-parameters come from a single-GPU fit on a sub-sample.

The tile arithmetic is behind a small engine interface so that the ownership / broadcast schedule can be
tested on CPU ranks (gloo) with a stub (tests/test_dist_gloo.py); ``HipTileEngine`` is the product engine and
needs the GPU -- there is no CPU fallback in the product path.
"""
import ctypes
import math

import numpy as np
import torch
import torch.distributed as dist

NB = 128
PANEL = 4            # 128-blocks per panel (OUTER_W of csrc/api.hip)
PW = NB * PANEL


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class Layout:
    """Who owns what: panel p (global block columns [4p, 4p+4)) lives on rank p % P at local panel slot p // P."""

    def __init__(self, n, rank, world):
        self.n = int(n)
        self.np = (self.n + NB - 1) // NB * NB
        self.nb = self.np // NB
        self.npanel = (self.nb + PANEL - 1) // PANEL
        self.rank, self.world = rank, world
        self.owned = [p for p in range(self.npanel) if p % world == rank]

    def owner(self, p):
        return p % self.world

    def width(self, p):
        """columns of panel p (the last one may be narrower)"""
        return min(PW, self.np - p * PW)

    def local_col0(self, p):
        """first local column of owned panel p"""
        return (p // self.world) * PW

    @property
    def local_cols(self):
        return max(1, len(self.owned)) * PW


class HipTileEngine:
    """The product engine: hand-written HIP behind the C ABI (include/gpimhip.h, gpimhip_dist_*)."""

    def __init__(self, layout, handle=None):
        from . import _lib
        self._lib = _lib
        self.H = handle or _lib.Handle()
        self.layout = layout
        self.device = self.H.device
        _lib.check(self.H.lib.gpimhip_dist_begin(self.H.h, layout.n))
        self.info = torch.zeros((4,), dtype=torch.int32, device=self.device)
        self.logdet = torch.zeros((layout.npanel, PANEL), dtype=torch.float64, device=self.device)

    def empty(self, rows, cols):
        return torch.zeros((rows, cols), dtype=torch.float64, device=self.device)

    def panel_factor(self, Aloc, p):
        L, lib = self.layout, self.H.lib
        self._lib.check(lib.gpimhip_dist_panel_factor(
            self.H.h, self._lib.ptr(Aloc), Aloc.stride(0), L.local_col0(p) // NB, p * PANEL,
            ctypes.c_void_p(self.logdet[p].data_ptr()), self._lib.ptr(self.info)))

    def trailing_update(self, panel, p, Aloc, c):
        L, lib = self.layout, self.H.lib
        self._lib.check(lib.gpimhip_dist_trailing_update(
            self.H.h, self._lib.ptr(panel), panel.stride(0), p * PANEL, self._lib.ptr(Aloc), Aloc.stride(0),
            L.local_col0(c) // NB, c * PANEL))

    def failed_column(self):
        return int(self.info[0].item())


class DistributedCholesky:
    """Block-column-cyclic right-looking Cholesky.  ``local`` is this rank's np x (512 * owned panels)
    share of the symmetric matrix (lower part meaningful, identity padding beyond n)."""

    def __init__(self, n, engine_factory=HipTileEngine, group=None):
        rank, world = _world()
        self.layout = Layout(n, rank, world)
        self.engine = engine_factory(self.layout)
        self.group = group
        self.local = self.engine.empty(self.layout.np, self.layout.local_cols)
        self._panel = [self.engine.empty(self.layout.np, PW) for _ in range(2)]

    # ------------------------------------------------------------------ filling the local share
    def set_from_function(self, cols_fn):
        """cols_fn(c0, c1) -> (n x (c1 - c0)) tensor with columns [c0, c1) of the matrix.  Padding: identity."""
        L = self.layout
        for p in L.owned:
            c0, c1 = p * PW, min(p * PW + PW, L.n)
            l0 = L.local_col0(p)
            if c1 > c0:
                self.local[:L.n, l0:l0 + (c1 - c0)] = cols_fn(c0, c1)
            for c in range(max(c0, L.n), p * PW + L.width(p)):        # identity padding
                self.local[c, l0 + (c - p * PW)] = 1.0
        return self

    # ------------------------------------------------------------------ factorisation
    def factor(self):
        L, eng = self.layout, self.engine
        for p in range(L.npanel):
            buf = self._panel[p & 1]
            w = L.width(p)
            if L.owner(p) == L.rank:
                eng.panel_factor(self.local, p)
                l0 = L.local_col0(p)
                buf[p * PW:, :w] = self.local[p * PW:, l0:l0 + w]
            if L.world > 1:
                dist.broadcast(buf, src=L.owner(p), group=self.group)
            # the next panel first: its owner can then factor and broadcast while the others still update
            todo = [c for c in L.owned if c > p]
            todo.sort(key=lambda c: (c != p + 1, c))
            for c in todo:
                eng.trailing_update(buf, p, self.local, c)
        bad = torch.tensor([eng.failed_column()], dtype=torch.int64)
        if L.world > 1:
            bad = bad.to(self.local.device if self.local.is_cuda else "cpu")
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
        if int(bad.item()) != 0:
            raise torch.linalg.LinAlgError("linalg.cholesky: the input is not positive-definite "
                                           "(leading minor of order %d)" % int(bad.item()))
        return self

    # ------------------------------------------------------------------ what the factor is for
    def logdet(self):
        """log det of the matrix = 2 sum log L_ii (every rank returns the same number)."""
        L = self.layout
        s = torch.zeros((1,), dtype=torch.float64, device=self.local.device)
        for p in L.owned:
            l0 = L.local_col0(p)
            w = L.width(p)
            d = torch.diagonal(self.local[p * PW:p * PW + w, l0:l0 + w])
            s += torch.log(d[:max(0, min(w, L.n - p * PW))]).sum()
        if L.world > 1:
            dist.all_reduce(s, group=self.group)
        return 2.0 * float(s.item())

    def solve(self, y):
        """alpha = (L L^T)^-1 y, replicated on every rank.  Two panel-by-panel triangular solves: O(N^2) work,
        one 512-double broadcast per panel each way (host-orchestrated torch ops on the local panels -- not
        part of the O(N^3) path)."""
        L = self.layout
        dev = self.local.device
        rhs_full = torch.zeros((L.np,), dtype=torch.float64, device=dev)
        rhs_full[:L.n] = torch.as_tensor(y, dtype=torch.float64).to(dev)
        # forward, z = L^-1 y.  Every rank accumulates acc = sum over ITS panels q of L(:, q) z_q; the rows of
        # panel p need the contributions of all ranks: one all-reduce of 512 doubles, then the owner solves
        # with the diagonal triangle and broadcasts z_p.
        z = torch.zeros((L.np,), dtype=torch.float64, device=dev)
        acc = torch.zeros((L.np,), dtype=torch.float64, device=dev)
        piece = torch.zeros((PW,), dtype=torch.float64, device=dev)
        for p in range(L.npanel):
            w, r0 = L.width(p), p * PW
            t = acc[r0:r0 + PW].clone() if r0 + PW <= L.np else torch.cat(
                [acc[r0:], torch.zeros((r0 + PW - L.np,), dtype=torch.float64, device=dev)])
            if L.world > 1:
                dist.all_reduce(t, group=self.group)
            if L.owner(p) == L.rank:
                l0 = L.local_col0(p)
                Lpp = torch.tril(self.local[r0:r0 + w, l0:l0 + w])
                piece[:w] = torch.linalg.solve_triangular(Lpp, (rhs_full[r0:r0 + w] - t[:w])[:, None], upper=False)[:, 0]
                acc[r0 + w:] += self.local[r0 + w:, l0:l0 + w] @ piece[:w]
            if L.world > 1:
                dist.broadcast(piece, src=L.owner(p), group=self.group)
            z[r0:r0 + w] = piece[:w]
        # backward, alpha = L^-T z: the owner of panel p holds L(rows below, panel p) and alpha of the rows
        # below is known to everybody by then
        a = torch.zeros((L.np,), dtype=torch.float64, device=dev)
        for p in reversed(range(L.npanel)):
            w, r0 = L.width(p), p * PW
            if L.owner(p) == L.rank:
                l0 = L.local_col0(p)
                Lpp = torch.tril(self.local[r0:r0 + w, l0:l0 + w])
                rhs = z[r0:r0 + w] - self.local[r0 + w:, l0:l0 + w].T @ a[r0 + w:]
                piece[:w] = torch.linalg.solve_triangular(Lpp.T, rhs[:, None], upper=True)[:, 0]
            if L.world > 1:
                dist.broadcast(piece, src=L.owner(p), group=self.group)
            a[r0:r0 + w] = piece[:w]
        self._z = z
        return a[:L.n]

    def solve_colsumsq(self, B):
        """q_j = |L^-1 B[:, j]|^2 for this rank's own right-hand sides B (np x m, rows beyond n zero; destroyed):
        the quadratic form of the posterior variance.  L is distributed by columns, the right-hand sides by
        rank, so the factor is streamed once more -- every owner re-broadcasts its panels in order -- and each
        rank forward-substitutes its columns: a 512-row triangular solve and a plain (np - r) x 512 x m GEMM
        per panel (rocBLAS through torch: ordinary library GEMMs, N^2 m flop per rank).  Collective: every
        rank must call it the same number of times."""
        L = self.layout
        q = torch.zeros((B.shape[1],), dtype=torch.float64, device=B.device)
        for p in range(L.npanel):
            buf = self._panel[p & 1]
            w, r0 = L.width(p), p * PW
            if L.owner(p) == L.rank:
                l0 = L.local_col0(p)
                buf[r0:, :w] = self.local[r0:, l0:l0 + w]
            if L.world > 1:
                dist.broadcast(buf, src=L.owner(p), group=self.group)
            if B.shape[1] == 0:
                continue
            Wp = torch.linalg.solve_triangular(torch.tril(buf[r0:r0 + w, :w]), B[r0:r0 + w], upper=False)
            q += (Wp * Wp).sum(0)
            if r0 + w < L.np:
                B[r0 + w:].addmm_(buf[r0 + w:, :w], Wp, alpha=-1.0)
        return q

    def nll(self, y):
        """1/2 y^T K^-1 y + 1/2 log det K + N/2 log 2 pi (no prior constant)."""
        alpha = self.solve(y)
        yv = torch.as_tensor(y, dtype=torch.float64).to(alpha.device)
        return 0.5 * float((yv * alpha).sum().item()) + 0.5 * self.logdet() + 0.5 * self.layout.n * math.log(2 * math.pi)

    def gather_lower(self):
        """The whole factor on every rank (tests and small problems only)."""
        L = self.layout
        full = torch.zeros((L.np, L.np), dtype=torch.float64, device=self.local.device)
        for p in range(L.npanel):
            w = L.width(p)
            buf = torch.zeros((L.np, PW), dtype=torch.float64, device=self.local.device)
            if L.owner(p) == L.rank:
                buf[:, :w] = self.local[:, L.local_col0(p):L.local_col0(p) + w]
            if L.world > 1:
                dist.broadcast(buf, src=L.owner(p), group=self.group)
            full[:, p * PW:p * PW + w] = buf[:, :w]
        return torch.tril(full)[:L.n, :L.n]


def exact_gp_posterior(X, y, Xtest, kernel="Matern52", lengthscale=None, variance=1.0, noise=1e-2,
                       jitter=1e-5, group=None, with_sd=True, chunk_bytes=1 << 32):
    """Posterior of ONE exact GP on all N points across the ranks of the process group, at fixed
    hyper-parameters (constrained values): what ``reconstructor.predict`` returns (gpr.py:247-250; mean, and sd
    with the noise included), for covariances too large for one device.  X (N, d), y (N,), Xtest (M, d) numpy /
    torch; every rank passes the same arrays and gets the same results back: (mean, sd, nll) -- sd is None
    when ``with_sd`` is False.  Test points are sharded over the ranks; the variance streams the factor
    through every rank once per ``chunk_bytes`` of K* columns (DistributedCholesky.solve_colsumsq)."""
    from . import _lib
    from .kernels import KernelSpec
    rank, world = _world()
    X = torch.as_tensor(X, dtype=torch.float64)
    N, d = X.shape
    ls = torch.as_tensor(lengthscale if lengthscale is not None else [1.0] * d, dtype=torch.float64).reshape(-1)
    spec = KernelSpec(kernel, d, [[0.0] * len(ls), (2 * ls).tolist()] if len(ls) > 1 else [0.0, float(2 * ls[0])])
    m = spec.struct()
    chol = DistributedCholesky(N, group=group)
    H = chol.engine.H
    dev = H.device
    Xd = X.to(dev).contiguous()
    theta = torch.cat([torch.tensor([variance], dtype=torch.float64), ls,
                       torch.ones(1, dtype=torch.float64)]).to(dev)

    def kmat(Z, out):
        _lib.check(H.lib.gpimhip_kmat(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(Z), Z.shape[0], _lib.ptr(theta),
                                      0.0, _lib.ptr(out), out.stride(0)))

    def cols(c0, c1):
        out = torch.empty((N, c1 - c0), dtype=torch.float64, device=dev)
        kmat(Xd[c0:c1].contiguous(), out)
        idx = torch.arange(c0, c1, device=dev)
        out[idx, idx - c0] += jitter + noise
        return out
    chol.set_from_function(cols).factor()
    alpha = chol.solve(y).contiguous()
    yv = torch.as_tensor(y, dtype=torch.float64).to(dev)
    nll = 0.5 * float((yv * alpha).sum().item()) + 0.5 * chol.logdet() + 0.5 * N * math.log(2 * math.pi)
    # test points sharded over the ranks, processed in chunks of K* columns
    Xt = torch.as_tensor(Xtest, dtype=torch.float64)
    M = Xt.shape[0]
    per = (M + world - 1) // world
    lo, hi = min(rank * per, M), min((rank + 1) * per, M)
    npd = chol.layout.np
    step = max(1, min(per, int(chunk_bytes // (8 * npd))))
    nchunk = (per + step - 1) // step                     # the same on every rank (collectives inside)
    part = torch.full((2, per), float("nan"), dtype=torch.float64, device=dev)
    for ci in range(nchunk):
        s0, s1 = min(hi, lo + ci * step), min(hi, lo + (ci + 1) * step)
        Ks = torch.zeros((npd, s1 - s0), dtype=torch.float64, device=dev)
        if s1 > s0:
            kmat(Xt[s0:s1].to(dev).contiguous(), Ks)
            part[0, s0 - lo:s1 - lo] = Ks[:N].T @ alpha
        if with_sd:
            q = chol.solve_colsumsq(Ks)
            part[1, s0 - lo:s1 - lo] = torch.sqrt(torch.clamp(variance - q, min=0.0) + noise)
    if world > 1:
        from .dist import all_gather
        parts = [torch.empty_like(part) for _ in range(world)]
        all_gather(parts, part, group=group)
        full = torch.cat([parts[r][:, :max(0, min(per, M - r * per))] for r in range(world)], dim=1)
    else:
        full = part[:, :M]
    res = full.cpu().numpy()
    return res[0], (res[1] if with_sd else None), nll


def exact_gp_posterior_mean(X, y, Xtest, **kw):
    """(mean, nll) only -- see exact_gp_posterior."""
    mean, _, nll = exact_gp_posterior(X, y, Xtest, with_sd=False, **kw)
    return mean, nll
