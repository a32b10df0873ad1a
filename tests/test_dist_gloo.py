"""world_size-2 gloo tests (CPU) of the unit sharding / gather / global top-k logic in
gpim_amd.dist.  The GP itself is replaced by a deterministic stub (the HIP engine has no CPU
path); what is covered is that every unit is computed exactly once, lands in the right slot on
rank 0, and that the arg-max reduction agrees with a single-process ranking."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from gpim_amd import dist as gd
    r, w, _ = gd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # --- independent units: 5 "slices" of 3x4; stub fit = simple functions of the data
    rng = np.random.default_rng(0)
    units = [rng.standard_normal((3, 4)) for _ in range(5)]
    calls = []

    def fit(R):
        calls.append(1)
        return R * 2.0, np.abs(R) + 1.0
    res = gd.run_units(units, fit, (3, 4), device=torch.device("cpu"))
    assert len(calls) == len(gd.shard_units(5, rank, world))
    if rank == 0:
        mean, sd = res
        for i, R in enumerate(units):
            np.testing.assert_array_equal(mean[i].numpy(), R * 2.0)
            np.testing.assert_array_equal(sd[i].numpy(), np.abs(R) + 1.0)
    else:
        assert res is None
    # --- candidate sharding + global top-k
    M, k = 103, 7
    acq = np.random.default_rng(1).standard_normal(M)
    acq[[5, 60]] = acq.max() + 1.0                   # a tie across the two ranks
    lo, hi = gd.candidate_block(M, rank, world)
    loc = acq[lo:hi]
    order = np.argsort(loc, kind="stable")[::-1][:k]
    lv = np.full(k, -np.inf)
    li = np.full(k, -1, dtype=np.int64)
    lv[:len(order)], li[:len(order)] = loc[order], order + lo
    gv, gi = gd.global_topk(lv, li, k)
    expect = np.argsort(acq, kind="stable")[::-1][:k]
    np.testing.assert_array_equal(gi.numpy(), expect)
    np.testing.assert_array_equal(gv.numpy(), acq[expect])
    # NaN-first ranking (the un-masked path of boptimizer.next_point) and block re-assembly
    acq2 = acq.copy()
    acq2[[3, 77]] = np.nan
    loc = acq2[lo:hi]
    order = np.argsort(loc, kind="stable")[::-1][:k]           # NaNs first after reversal
    lv = np.full(k, -np.inf)
    li = np.full(k, -1, dtype=np.int64)
    lv[:len(order)], li[:len(order)] = loc[order], order + lo
    gv, gi = gd.global_topk(lv, li, k, nan_first=True)
    expect = np.argsort(acq2, kind="stable")[::-1][:k]
    np.testing.assert_array_equal(gi.numpy(), expect)
    full = gd.all_gather_blocks(torch.from_numpy(acq[lo:hi].copy()), M)
    np.testing.assert_array_equal(full.numpy(), acq)
    ret[rank] = True
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get(0) and ret.get(1)


def test_single_process_paths():
    from gpim_amd import dist as gd
    assert gd.world() == (0, 1)
    assert gd.shard_units(5) == [0, 1, 2, 3, 4]
    assert gd.shard_units(7, 1, 3) == [1, 4]
    assert gd.candidate_block(10, 2, 4) == (6, 9) and gd.candidate_block(10, 3, 4) == (9, 10)
    units = [np.full((2, 2), float(i)) for i in range(3)]
    mean, sd = gd.run_units(units, lambda R: (R + 1, R + 2), (2, 2), device=torch.device("cpu"))
    assert mean.shape == (3, 2, 2) and sd[2, 0, 0].item() == 4.0


# ------------------------------------------------------------------------------------------------
# block-column-cyclic Cholesky: ownership / broadcast schedule with a stubbed tile engine
# ------------------------------------------------------------------------------------------------
class NumpyTileEngine:
    """Test stub of gpim_amd.dist_chol.HipTileEngine (the product engine needs the GPU): the same tile operations
    in plain torch-CPU arithmetic, so that world-size-2 gloo ranks can run the real schedule (no side stream: the
    driver falls back to in-order calls)."""

    def __init__(self, layout):
        self.layout = layout
        self.bad = 0

    def empty(self, rows, cols):
        return torch.zeros((rows, cols), dtype=torch.float64)

    def panel_factor(self, Aloc, p):
        from gpim_amd.dist_chol import PW
        L = self.layout
        self._Aloc = Aloc
        w, r0, l0 = L.width(p), p * PW, L.local_col0(p)
        A = Aloc[r0:r0 + w, l0:l0 + w]
        sym = torch.tril(A) + torch.tril(A, -1).T
        Lpp, info = torch.linalg.cholesky_ex(sym)
        if int(info) != 0 and self.bad == 0:
            self.bad = r0 + int(info)
        Aloc[r0:r0 + w, l0:l0 + w] = Lpp
        if r0 + w < L.np:
            Aloc[r0 + w:, l0:l0 + w] = torch.linalg.solve_triangular(Lpp, Aloc[r0 + w:, l0:l0 + w].T, upper=False).T

    def pack(self, Aloc, p, buf):
        from gpim_amd.dist_chol import PW
        L = self.layout
        w, r0, l0 = L.width(p), p * PW, L.local_col0(p)
        buf[r0:L.np, :w] = Aloc[r0:, l0:l0 + w]

    def update(self, panel, p, Aloc, first, last):
        from gpim_amd.dist_chol import PW
        L = self.layout
        wp = L.width(p)
        for c in L.owned:
            if first <= c < last:
                wc, c0, l0 = L.width(c), c * PW, L.local_col0(c)
                Aloc[c0:, l0:l0 + wc] -= panel[c0:L.np, :wp] @ panel[c0:c0 + wc, :wp].T

    def solve_update(self, buf, p, B, Wt, q, col_tiles=0):
        from gpim_amd.dist_chol import PW
        L = self.layout
        w, r0 = L.width(p), p * PW
        nc = col_tiles * 128 if col_tiles else B.shape[1]
        Wp = torch.linalg.solve_triangular(torch.tril(buf[r0:r0 + w, :w]), B[r0:r0 + w, :nc], upper=False)
        Wt[:w, :nc] = Wp
        if q is not None:
            q[:nc] += (Wp * Wp).sum(0)
        if r0 + w < L.np:
            B[r0 + w:, :nc].addmm_(buf[r0 + w:L.np, :w], Wp, alpha=-1.0)

    def solve_update2(self, wide, p, B, Wt2, q, col_tiles, second):
        """Panels in pairs (gpimhip_dist_solve_update2): the first updates the second panel's rows only, the rows below the
        pair receive both panels at once."""
        from gpim_amd.dist_chol import PW
        L = self.layout
        w, r0 = L.width(p), p * PW
        nc = col_tiles * 128 if col_tiles else B.shape[1]
        half = wide[:, second * PW:(second + 1) * PW]
        Wp = torch.linalg.solve_triangular(torch.tril(half[r0:r0 + w, :w]), B[r0:r0 + w, :nc], upper=False)
        Wt2[second * PW:second * PW + w, :nc] = Wp
        if q is not None:
            q[:nc] += (Wp * Wp).sum(0)
        if not second:
            r1 = min(L.np, r0 + w + PW)
            if r0 + w < L.np:
                B[r0 + w:r1, :nc].addmm_(half[r0 + w:r1, :w], Wp, alpha=-1.0)
        elif r0 + w < L.np:
            B[r0 + w:, :nc].addmm_(wide[r0 + w:L.np, :PW + w], Wt2[:PW + w, :nc], alpha=-1.0)

    def kinv_update(self, xbuf, c, Xloc, Kinv):
        from gpim_amd.dist_chol import PW
        L = self.layout
        w, r0 = L.width(c), c * PW
        # rows of panel c of X^T X against the owned columns (X is zero above its diagonal blocks: only rows >= r0 travel)
        full = xbuf[r0:L.np, :w].T @ Xloc[r0:]
        for p in L.owned:
            if p <= c:                                          # block columns left of / at the panel: tiles i >= j
                l0 = L.local_col0(p)
                Kinv[r0:r0 + w, l0:l0 + L.width(p)] = full[:, l0:l0 + L.width(p)]

    def vec_forward(self, Aloc, p, y_p, t, piece, acc):
        from gpim_amd.dist_chol import PW
        L = self.layout
        w, r0, l0 = L.width(p), p * PW, L.local_col0(p)
        Lpp = torch.tril(Aloc[r0:r0 + w, l0:l0 + w])
        piece[:w] = torch.linalg.solve_triangular(Lpp, (y_p[:w] - t[:w])[:, None], upper=False)[:, 0]
        acc[r0 + w:L.np] += Aloc[r0 + w:, l0:l0 + w] @ piece[:w]

    def vec_backward(self, Aloc, p, z_p, a, work, piece):
        from gpim_amd.dist_chol import PW
        L = self.layout
        w, r0, l0 = L.width(p), p * PW, L.local_col0(p)
        Lpp = torch.tril(Aloc[r0:r0 + w, l0:l0 + w])
        rhs = z_p[:w] - Aloc[r0 + w:, l0:l0 + w].T @ a[r0 + w:L.np]
        piece[:w] = torch.linalg.solve_triangular(Lpp.T, rhs[:, None], upper=True)[:, 0]

    def matvec_t(self, A, x, out):
        out[:A.shape[1]] = A.T @ x

    def half_logdet_owned(self):
        from gpim_amd.dist_chol import PW
        L = self.layout
        s = torch.zeros((1,), dtype=torch.float64)
        for p in L.owned:
            d = torch.diagonal(self._Aloc[p * PW:p * PW + L.width(p), L.local_col0(p):L.local_col0(p) + L.width(p)])
            s += torch.log(d).sum()
        return s

    def failed_column(self):
        return self.bad


def _chol_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from gpim_amd import dist as gd
    from gpim_amd.dist_chol import DistributedCholesky, Layout
    gd.init_from_env(backend="gloo")
    for n in (1500, 512, 700):                       # 3 panels (ragged last), 1 panel, 2 panels
        rng = np.random.default_rng(n)
        B = rng.standard_normal((n, n // 2))
        A = torch.from_numpy(B @ B.T + n * np.eye(n))
        y = torch.from_numpy(rng.standard_normal(n))
        ch = DistributedCholesky(n, engine_factory=NumpyTileEngine)
        ch._wide = [ch.engine.empty(ch.layout.np + 128, 1024) for _ in range(2)]
        for b in ch._panel + ch._wide:
            b.fill_(float("nan"))        # a consumer that read rows which no broadcast delivers would spread NaNs
        lay = ch.layout
        assert lay.owned == [p for p in range(lay.npanel) if p % world == rank]
        assert ch.local.shape == (lay.np, max(1, len(lay.owned)) * 512)
        ch.set_from_function(lambda c0, c1: A[:, c0:c1]).factor()
        Lref = torch.linalg.cholesky(A)
        Lfull = ch.gather_lower()
        assert torch.allclose(Lfull, Lref, rtol=1e-11, atol=1e-11)
        assert abs(ch.logdet() - torch.logdet(A).item()) < 1e-9 * abs(torch.logdet(A).item())
        alpha = ch.solve(y)
        assert torch.allclose(alpha, torch.cholesky_solve(y[:, None], Lref)[:, 0], rtol=1e-9, atol=1e-12)
        ref_nll = 0.5 * float(y @ torch.cholesky_solve(y[:, None], Lref)[:, 0]) + 0.5 * torch.logdet(A).item() \
            + 0.5 * n * np.log(2 * np.pi)
        assert abs(ch.nll(y) - ref_nll) < 1e-9 * abs(ref_nll)
        # variance quadratic form for rank-local right-hand sides (different widths per rank, incl. none)
        m_loc = [7, 0, 3][rank]
        Bm = torch.zeros((lay.np, m_loc), dtype=torch.float64)
        Bm[:n] = torch.from_numpy(np.random.default_rng(100 + rank).standard_normal((n, m_loc)))
        want = (torch.linalg.solve_triangular(Lref, Bm[:n].clone(), upper=False) ** 2).sum(0) if m_loc else torch.zeros(0)
        got = ch.solve_colsumsq(Bm)
        assert torch.allclose(got, want.to(torch.float64), rtol=1e-9, atol=1e-12)
        # training passes: X = L^-1 and K^-1 = X^T X, both distributed like L (lower parts against torch)
        Xl = ch.inverse()
        Kl = ch.kinv(Xl)
        Xref = torch.linalg.inv(Lref)
        Kref = torch.linalg.inv(A)
        for p in lay.owned:
            c0, w, l0 = p * 512, min(512, n - p * 512), lay.local_col0(p)
            assert torch.allclose(Xl[:n, l0:l0 + w], Xref[:, c0:c0 + w], rtol=1e-9, atol=1e-12)
            mask = (torch.arange(n)[:, None] >= torch.arange(c0, c0 + w)[None, :])
            assert torch.allclose(Kl[:n, l0:l0 + w][mask], Kref[:, c0:c0 + w][mask], rtol=1e-9, atol=1e-12)
    # a matrix that is not positive-definite is reported on every rank
    A = torch.eye(700, dtype=torch.float64)
    A[600, 600] = -1.0
    ch = DistributedCholesky(700, engine_factory=NumpyTileEngine)
    try:
        ch.set_from_function(lambda c0, c1: A[:, c0:c1]).factor()
        raised = False
    except torch.linalg.LinAlgError:
        raised = True
    assert raised
    ret[rank] = True
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_block_cyclic_cholesky_schedule_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_chol_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world))


def test_block_cyclic_layout():
    from gpim_amd.dist_chol import Layout
    lay = Layout(65536, 3, 8)
    assert lay.nb == 512 and lay.npanel == 128 and lay.owned == list(range(3, 128, 8))
    assert lay.owner(19) == 3 and lay.local_col0(19) == 2 * 512 and lay.local_cols == 16 * 512
    lay = Layout(700, 0, 1)
    assert lay.np == 768 and lay.npanel == 2 and lay.width(1) == 256 and lay.owned == [0, 1]
