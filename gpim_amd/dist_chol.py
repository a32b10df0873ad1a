"""
dist_chol.py -- ONE exact GP across the GPUs of a node: block-column-cyclic Cholesky of the covariance
over ``torch.distributed`` (RCCL over xGMI), SURVEY 8(f) rank 1.

The reference fits a whole cube as one d-dimensional GP (gpim/gpreg/gpr.py:30-43,115-126); its dense
covariance stops fitting one device beyond N ~ 10^5 (N = 65536, the complete 256 x 256 image of config
C2, is 32 GiB).  Here the N x N matrix is dealt to the P ranks by 512-column panels -- panel p belongs to
rank p mod P, the 1 x P case of a 2-D block-cyclic layout: every rank owns whole columns, so a panel is
factored by its owner alone and ONE broadcast per panel (np x 512 doubles, 268 MB at N = 65536) is the
only data-path collective.  Round p:

    owner(p) : factor panel p in place              (C ABI: gpimhip_dist_panel_factor -- the step launches of
                                                      csrc/cholstep.hip restricted to the panel)
    all      : broadcast of the factored panel       (dist.broadcast, src = owner(p))
    each     : trailing update of every owned panel right of p with the broadcast copy, ONE launch
                                                     (gpimhip_dist_update -- the fp64 MFMA tile engine)

Look-ahead: the owner of panel p+1 updates that panel FIRST, then factors it, packs it and starts its
broadcast on a high-priority SIDE stream while the main stream goes on with the rest of round p's updates (all of a
rank's panels in ONE launch of the tile engine); the broadcast is asynchronous (``async_op=True``) and the main
stream only waits for it at the top of round p+1.  The broadcast buffer carries the panel's rows and, in 128 extra
rows, the inverses of its diagonal blocks (the forward substitutions of the other ranks multiply by them).
Per-rank memory: N^2 / P doubles + two panel buffers.  Communication per rank: N^2 / 2 doubles received in
total, about the time of the 1/P share of the N^3 / 3 flop at N = 65536, P = 8.

What the distributed model offers: the factor, log det K, the negative log marginal likelihood, the posterior mean
and standard deviation, and TRAINING (``exact_gp_fit``: distributed L^-1 and K^-1 = X^T X for the gradient, one
all-reduce of 8 doubles per Adam iteration).  alpha = K^-1 y is two panel-wise triangular solves on the owners
(gpimhip_dist_vec_forward / _backward: block substitution with the diagonal-block inverses + HBM-bound mat-vecs over
the panel's rows, O(N^2)); K*^T alpha is sharded over test points (gpimhip_matvec_t); for the variance, the inverse
and K^-1 the factor (or X) is streamed through the ranks once more -- the next panel's broadcast is in flight
(``async_op=True``, two buffers) while the current one is consumed on the MFMA tile engine (gpimhip_dist_solve_update,
gpimhip_dist_kinv_update).

The tile arithmetic is behind a small engine interface so that the ownership / broadcast schedule can be
tested on CPU ranks (gloo) with a stub (tests/test_dist_gloo.py); ``HipTileEngine`` is the product engine and
needs the GPU -- there is no CPU fallback in the product path.
"""
import contextlib
import ctypes
import math
import os

import numpy as np
import torch
import torch.distributed as dist

NB = 128
PANEL = 4            # 128-blocks per panel (OUTER_W of csrc/api.hip)
PW = NB * PANEL
KINV_GROUP = int(os.environ.get("GPIM_DIST_KINV_GROUP", "8"))      # panels per launch of the K^-1 pass (1: one at a time)


def _force_collectives():
    """Test mode (GPIM_DIST_FORCE_COLLECTIVES=1): issue every broadcast / all-reduce of the schedule at world size 1 too.
    A one-rank RCCL communicator is legal, so a one-GPU box can run the collectives' REAL stream semantics (the
    collective on RCCL's own stream, ``async_op=True`` Work handles consumed on another stream), which gloo -- blocking the
    host -- cannot show (tests/tools/nccl1_worker.py)."""
    return os.environ.get("GPIM_DIST_FORCE_COLLECTIVES", "0") not in ("", "0")


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


class _Ready:
    """Stands in for the Work handle of an asynchronous broadcast when there is nothing to wait for."""

    def wait(self):
        return True


class HipTileEngine:
    """The product engine: hand-written HIP behind the C ABI (include/gpimhip.h, gpimhip_dist_*).  Two library
    handles: one on the caller's (main) stream for the updates and solves, one on a high-priority side stream
    for the panel chain, so that factoring panel p+1 overlaps the rest of round p's updates."""

    grouped_kinv = True          # kinv_update takes several panels side by side (DistributedCholesky.kinv)

    def __init__(self, layout, handle=None):
        from . import _lib
        self._lib = _lib
        self.H = handle or _lib.Handle()
        self.layout = layout
        self.device = self.H.device
        self.main = torch.cuda.current_stream(self.device)
        self.side = torch.cuda.Stream(device=self.device, priority=-1)
        self.Hs = _lib.Handle(stream=self.side)
        _lib.check(self.H.lib.gpimhip_dist_setup(self.H.h, layout.n, layout.world, layout.rank))
        _lib.check(self.H.lib.gpimhip_dist_setup(self.Hs.h, layout.n, layout.world, layout.rank))
        self.info = torch.zeros((4,), dtype=torch.int32, device=self.device)
        self.logdet = torch.zeros((layout.npanel, PANEL), dtype=torch.float64, device=self.device)
        self._ev = torch.cuda.Event()
        self._own = None
        # fault injection (tests/tools/dist2_worker.py): GPIM_DIST_FAULT_DELAY_US > 0 puts a spin kernel of that length in
        # front of EVERY engine launch, on the stream the launch goes to -- producers on the side stream (panel chain, pack,
        # vector solves) and consumers on the main stream alike.  Every ordering between the two streams and the broadcasts
        # is an explicit event / Work edge; with the launches pushed apart, a missing edge reads or overwrites a buffer
        # at the wrong time and the result differs from the undelayed run.
        self._delay_cycles = int(float(os.environ.get("GPIM_DIST_FAULT_DELAY_US", "0")) * 2000)

    def _inject(self, side):
        if self._delay_cycles:
            with torch.cuda.stream(self.side if side else torch.cuda.current_stream(self.device)):
                torch.cuda._sleep(self._delay_cycles)

    def empty(self, rows, cols):
        return torch.zeros((rows, cols), dtype=torch.float64, device=self.device)

    # ---- the panel chain (side stream)
    def side_stream(self):
        """Context in which panel_factor / pack / the broadcast of the packed panel are issued: the side stream,
        after everything the main stream has been given so far."""
        self._ev.record(self.main)
        self.side.wait_event(self._ev)
        return torch.cuda.stream(self.side)

    def side_done(self):
        """Something with .wait(): makes the stream current at that time wait for the side stream (P = 1: no
        broadcast whose Work handle would do that)."""
        ev = torch.cuda.Event()
        ev.record(self.side)

        class _W:
            def wait(_self):
                torch.cuda.current_stream(self.device).wait_event(ev)
                return True
        return _W()

    def panel_factor(self, Aloc, p):
        self._inject(True)
        L, lib = self.layout, self.Hs.lib
        self._lib.check(lib.gpimhip_dist_panel_factor(
            self.Hs.h, self._lib.ptr(Aloc), Aloc.stride(0), L.local_col0(p) // NB, p * PANEL,
            ctypes.c_void_p(self.logdet[p].data_ptr()), self._lib.ptr(self.info)))

    def pack(self, Aloc, p, buf):
        self._inject(True)
        L, lib = self.layout, self.Hs.lib
        self._lib.check(lib.gpimhip_dist_panel_pack(
            self.Hs.h, self._lib.ptr(Aloc), Aloc.stride(0), L.local_col0(p) // NB, p * PANEL, self._lib.ptr(buf),
            buf.stride(0)))

    # ---- main stream
    def update(self, buf, p, Aloc, first, last):
        """Trailing update of the owned panels c, first <= c < last, with the packed panel p: one launch."""
        self._inject(False)
        lib = self.H.lib
        self._lib.check(lib.gpimhip_dist_update(self.H.h, self._lib.ptr(buf), buf.stride(0), p * PANEL,
                                                self._lib.ptr(Aloc), Aloc.stride(0), first, last))

    def solve_update(self, buf, p, B, Wt, q, col_tiles=0):
        self._inject(False)
        lib = self.H.lib
        self._lib.check(lib.gpimhip_dist_solve_update(self.H.h, self._lib.ptr(buf), buf.stride(0), p * PANEL,
                                                      self._lib.ptr(B), B.stride(0), B.shape[1], self._lib.ptr(Wt),
                                                      Wt.stride(0), self._lib.ptr(q), col_tiles))

    def solve_update2(self, wide, p, B, Wt2, q, col_tiles, second):
        self._inject(False)
        lib = self.H.lib
        self._lib.check(lib.gpimhip_dist_solve_update2(self.H.h, self._lib.ptr(wide), wide.stride(0), p * PANEL,
                                                       self._lib.ptr(B), B.stride(0), B.shape[1], self._lib.ptr(Wt2),
                                                       Wt2.stride(0), self._lib.ptr(q), col_tiles, int(second)))

    def kinv_update(self, xbuf, c, Xloc, Kinv, npanels=1):
        """Rows of the panels c ... c + npanels - 1 of K^-1 = X^T X (xbuf: those block columns of X side by side) against the
        owned columns: one launch."""
        self._inject(False)
        lib = self.H.lib
        self._lib.check(lib.gpimhip_dist_kinv_update_n(self.H.h, self._lib.ptr(xbuf), xbuf.stride(0), c * PANEL, int(npanels),
                                                       self._lib.ptr(Xloc), Xloc.stride(0), self._lib.ptr(Kinv),
                                                       Kinv.stride(0)))

    # ---- O(N^2) vector solves on the owner of a panel: the side handle factored it and holds its diagonal-block inverses
    def vec_forward(self, Aloc, p, y_p, t, piece, acc):
        self._inject(True)
        L, lib = self.layout, self.Hs.lib
        self._lib.check(lib.gpimhip_dist_vec_forward(self.Hs.h, self._lib.ptr(Aloc), Aloc.stride(0), L.local_col0(p) // NB,
                                                     p * PANEL, ctypes.c_void_p(y_p.data_ptr()),
                                                     ctypes.c_void_p(t.data_ptr()), self._lib.ptr(piece), self._lib.ptr(acc)))

    def vec_backward(self, Aloc, p, z_p, a, work, piece):
        self._inject(True)
        L, lib = self.layout, self.Hs.lib
        self._lib.check(lib.gpimhip_dist_vec_backward(self.Hs.h, self._lib.ptr(Aloc), Aloc.stride(0), L.local_col0(p) // NB,
                                                      p * PANEL, ctypes.c_void_p(z_p.data_ptr()), self._lib.ptr(a),
                                                      self._lib.ptr(work), self._lib.ptr(piece)))

    def matvec_t(self, A, x, out):
        """out = A^T x (A: rows x cols view of a row-major matrix, cols a multiple of 64)"""
        self._lib.check(self.H.lib.gpimhip_matvec_t(self.H.h, ctypes.c_void_p(A.data_ptr()), A.stride(0), A.shape[0],
                                                    A.shape[1], ctypes.c_void_p(x.data_ptr()), self._lib.ptr(out)))

    def half_logdet_owned(self):
        """sum of log L_ii over the owned panels (the factorisation role's per-block partial sums; padding rows are
        identity: log 1 = 0) as a 1-element device tensor"""
        if self._own is None:
            self._own = torch.tensor(self.layout.owned, dtype=torch.long, device=self.device)
        return self.logdet[self._own].sum().reshape(1) if len(self.layout.owned) else torch.zeros((1,), dtype=torch.float64,
                                                                                                 device=self.device)

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
        self.local = self._matrix(self.layout.np, self.layout.local_cols)
        # broadcast buffers: the panel's rows + 128 rows of diagonal-block inverses
        self._panel = [self.engine.empty(self.layout.np + NB, PW) for _ in range(2)]
        self._X = self._Wt = None                      # workspaces of ``inverse``
        self._wide = None                              # two panels side by side (``_stream_pairs``), two such buffers
        self._group = None                             # KINV_GROUP panels side by side (``_stream_groups``), two such buffers
        self._factored = False                         # ``local`` holds a factor (set by factor(), cleared by kinv(out=local))
        self._coll = self.layout.world > 1 or (_force_collectives() and dist.is_available() and dist.is_initialized())

    def _need_factor(self, what):
        """``kinv(X, out=self.local)`` writes K^-1's lower tiles over the factor (tiles above the diagonal keep what
        they held): every consumer of the factor refuses to run on that."""
        if not self._factored:
            raise RuntimeError("DistributedCholesky.%s: `local` does not hold a factor (call factor() first; "
                               "kinv(..., out=self.local) overwrites it)" % what)

    def _matrix(self, rows, cols):
        """A rows x cols workspace matrix whose leading dimension is NOT a large power of two: with 512 * owned-panels
        columns (65536 at N = 65536 on one GPU) the k-rows of an operand tile are 512 KB apart and land in few HBM
        channels; 256 extra columns spread them."""
        pad = 256 if cols % 2048 == 0 else 0
        return self.engine.empty(rows, cols + pad)[:, :cols]

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
    def _factor_and_send(self, p):
        """Owner of panel p: factor, pack and start the broadcast -- on the engine's side stream.  Others: post the
        receive.  Returns something with .wait() that orders the caller's stream behind the arrival of the panel."""
        L, eng = self.layout, self.engine
        buf = self._panel[p & 1]
        if L.owner(p) == L.rank:
            with self._side():
                eng.panel_factor(self.local, p)
                eng.pack(self.local, p, buf)
                if self._coll:
                    return dist.broadcast(self._wire(buf, p), src=L.rank, group=self.group, async_op=True)
            return self._side_done()
        return dist.broadcast(self._wire(buf, p), src=L.owner(p), group=self.group, async_op=True)

    @staticmethod
    def _wire(buf, p):
        """What travels for panel p: rows p * 512 ... of the buffer (contiguous: the panel's rows from its diagonal block
        down, then the 128 rows of diagonal-block inverses).  The rows above are structural zeros of L and of X = L^-1
        alike and no consumer reads them (gpimhip_dist_update: tiles right of the panel; gpimhip_dist_solve_update: block
        rows >= 4 p; gpimhip_dist_kinv_update: k-blocks >= the tile's row) -- on a receiver they hold whatever an earlier
        panel left there.  Halves the volume: sum_p (np - 512 p + 128) * 512 doubles = N^2 / 2 + ... instead of N^2."""
        return buf[p * PW:]

    def _side(self):
        eng = self.engine
        return eng.side_stream() if hasattr(eng, "side_stream") else contextlib.nullcontext()

    def _side_done(self):
        eng = self.engine
        return eng.side_done() if hasattr(eng, "side_done") else _Ready()

    def factor(self, check=True):
        """check=False: no host synchronisation -- the caller reads the failure flag itself (``failed_flag``)."""
        L, eng = self.layout, self.engine
        work = self._factor_and_send(0)
        for p in range(L.npanel):
            buf = self._panel[p & 1]
            work.wait()                                  # the main stream waits for panel p
            nxt = p + 1
            rest = nxt
            if nxt < L.npanel:
                if L.owner(nxt) == L.rank:
                    # look-ahead: bring panel p+1 up to date first; its chain and broadcast then run on the side
                    # stream beside the remaining updates of this round
                    eng.update(buf, p, self.local, nxt, nxt + 1)
                    rest = nxt + 1
                work = self._factor_and_send(nxt)
            if rest < L.npanel:
                eng.update(buf, p, self.local, rest, L.npanel)
        self._factored = True
        if not check:
            return self
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
        self._need_factor("logdet")
        s = self.engine.half_logdet_owned().clone()
        if self.layout.world > 1:
            dist.all_reduce(s, group=self.group)
        return 2.0 * float(s.item())

    def solve(self, y):
        """alpha = (L L^T)^-1 y, replicated on every rank.  Two panel-by-panel triangular solves, O(N^2) work: the owner
        of a panel solves with its diagonal triangle and multiplies the rows below (engine.vec_forward / vec_backward);
        per panel one all-reduce of the 512 partial sums (forward only) and one broadcast of the 512 solved entries."""
        self._need_factor("solve")
        L, eng = self.layout, self.engine
        dev = self.local.device
        rhs_full = torch.zeros((L.npanel * PW,), dtype=torch.float64, device=dev)
        rhs_full[:L.n] = torch.as_tensor(y, dtype=torch.float64).to(dev)
        # forward, z = L^-1 y.  Every rank accumulates acc = sum over ITS panels q of L(:, q) z_q; the rows of
        # panel p need the contributions of all ranks: one all-reduce of 512 doubles, then the owner solves
        # with the diagonal triangle and broadcasts z_p.
        z = torch.zeros((L.npanel * PW,), dtype=torch.float64, device=dev)
        acc = torch.zeros((L.npanel * PW,), dtype=torch.float64, device=dev)
        piece = torch.zeros((PW,), dtype=torch.float64, device=dev)
        work = torch.zeros((PW,), dtype=torch.float64, device=dev)
        t = torch.zeros((PW,), dtype=torch.float64, device=dev)
        for p in range(L.npanel):
            r0 = p * PW
            t.copy_(acc[r0:r0 + PW])
            if self._coll:
                dist.all_reduce(t, group=self.group)
            if L.owner(p) == L.rank:
                with self._side():
                    eng.vec_forward(self.local, p, rhs_full[r0:r0 + PW], t, piece, acc)
                self._side_done().wait()
            if self._coll:
                dist.broadcast(piece, src=L.owner(p), group=self.group)
            z[r0:r0 + PW].copy_(piece)
        # backward, alpha = L^-T z: the owner of panel p holds L(rows below, panel p) and alpha of the rows
        # below is known to everybody by then
        a = torch.zeros((L.npanel * PW,), dtype=torch.float64, device=dev)
        for p in reversed(range(L.npanel)):
            r0 = p * PW
            if L.owner(p) == L.rank:
                with self._side():
                    eng.vec_backward(self.local, p, z[r0:r0 + PW], a, work, piece)
                self._side_done().wait()
            if self._coll:
                dist.broadcast(piece, src=L.owner(p), group=self.group)
            a[r0:r0 + PW].copy_(piece)
        self._z = z[:L.np]
        return a[:L.n]

    def solve_colsumsq(self, B):
        """q_j = |L^-1 B[:, j]|^2 for this rank's own right-hand sides B (np x m, rows beyond n zero; destroyed):
        the quadratic form of the posterior variance.  L is distributed by columns, the right-hand sides by
        rank, so the factor is streamed once more -- every owner re-broadcasts its packed panels in order -- and
        each rank forward-substitutes its columns panel by panel on the MFMA tile engine
        (gpimhip_dist_solve_update: block solves with the diagonal-block inverses, a k-depth-512 update of the
        rows below, column sums of squares; N^2 m flop per rank).  Collective: every rank must call it the same
        number of times.  The engine works on column counts that are multiples of 128: B is padded if needed."""
        L, eng = self.layout, self.engine
        m = B.shape[1]
        mpad = max(NB, (m + NB - 1) // NB * NB)
        if B.shape[1] != mpad or not B.is_contiguous():
            Bp = eng.empty(L.np, mpad)
            Bp[:, :m] = B
            B = Bp
        Wt2 = eng.empty(2 * PW, mpad)
        q = torch.zeros((mpad,), dtype=torch.float64, device=B.device)
        for p, wide, second in self._stream_factor_pairs():
            if m:
                eng.solve_update2(wide, p, B, Wt2, q, 0, second)
        return q[:m]

    def _start_panel(self, p, fill):
        """Owner of panel p: fill(buf) -- on the engine's side stream, behind everything the main stream has been given
        so far (the previous consumer of this buffer) -- and start the broadcast; others: post the receive.  Returns
        something with .wait() that orders the caller's stream behind the arrival of the buffer."""
        L = self.layout
        buf = self._panel[p & 1]
        if L.owner(p) == L.rank:
            with self._side():
                fill(buf)
                if self._coll:
                    return dist.broadcast(self._wire(buf, p), src=L.rank, group=self.group, async_op=True)
            return self._side_done()
        return dist.broadcast(self._wire(buf, p), src=L.owner(p), group=self.group, async_op=True)

    def _stream(self, fill_of):
        """Yields (p, buffer) for every panel in order; the broadcast of panel p + 1 is in flight while the caller
        consumes panel p (two buffers)."""
        L = self.layout
        work = self._start_panel(0, fill_of(0))
        for p in range(L.npanel):
            work.wait()
            if p + 1 < L.npanel:
                work = self._start_panel(p + 1, fill_of(p + 1))
            yield p, self._panel[p & 1]

    def _stream_factor(self):
        """The factored panels once more (packed with the inverses of their diagonal blocks)."""
        self._need_factor("_stream_factor")
        return self._stream(lambda p: (lambda buf: self.engine.pack(self.local, p, buf)))

    def _stream_pairs(self, fill_of):
        """Yields (p, wide, second): the panels in order, each copied from its (contiguous) broadcast buffer into one half
        of a buffer of 1024 columns -- panel 2q in columns [0, 512) and panel 2q + 1 in [512, 1024) of buffer q & 1 -- so
        that a consumer can apply both panels of a pair in one pass of k-depth 1024 (engine.solve_update2).  The copy is
        268 MB at N = 65536, a tenth of a millisecond per panel."""
        L = self.layout
        if self._wide is None:
            self._wide = [self.engine.empty(L.np + NB, 2 * PW) for _ in range(2)]
        for p, buf in self._stream(fill_of):
            wide, second = self._wide[(p // 2) & 1], p & 1
            wide[p * PW:, second * PW:(second + 1) * PW].copy_(buf[p * PW:])
            yield p, wide, second

    def _stream_groups(self, fill_of, G):
        """Yields (p0, wide, count): the panels in order, G at a time side by side in one buffer of G * 512 columns (two such
        buffers take turns), copied from their broadcast buffers as they arrive -- the next panel's broadcast is in flight
        meanwhile, as in ``_stream``.  The last group may be shorter."""
        L = self.layout
        if self._group is None or self._group[0].shape[1] != G * PW:
            self._group = [self.engine.empty(L.np + NB, G * PW) for _ in range(2)]
        for p, buf in self._stream(fill_of):
            g, slot = p // G, p % G
            wide = self._group[g & 1]
            wide[p * PW:, slot * PW:(slot + 1) * PW].copy_(buf[p * PW:])
            if slot == G - 1 or p == L.npanel - 1:
                yield g * G, wide, slot + 1

    def _stream_factor_pairs(self):
        self._need_factor("_stream_factor_pairs")
        return self._stream_pairs(lambda p: (lambda buf: self.engine.pack(self.local, p, buf)))

    def inverse(self):
        """X = L^-1, distributed like L: this rank's block columns (np x 512 * owned panels), lower triangular.
        The factor is streamed through the ranks once more and every rank forward-substitutes the identity
        columns it owns (gpimhip_dist_solve_update; block columns right of the current panel are still zero and
        are skipped, so the work is that of a triangular inversion, N^3 / 3 flop over all ranks).
        In place: block row p of the right-hand sides is dead once step p has turned it into block row p of X, which
        takes its place (through the 512-row staging buffer the engine writes).  The matrix is a persistent workspace
        of this object, re-seeded with the identity on every call -- the returned tensor is valid until the next call."""
        L, eng = self.layout, self.engine
        if self._X is None:
            self._X = self._matrix(L.np, L.local_cols)
            self._Wt = self._matrix(2 * PW, L.local_cols)
        else:
            self._X.zero_()
        X, Wt2 = self._X, self._Wt
        for p in L.owned:
            idx = torch.arange(L.width(p), device=X.device)
            X[p * PW + idx, L.local_col0(p) + idx] = 1.0
        nprev = 0
        for p, wide, second in self._stream_factor_pairs():
            nown = sum(1 for c in L.owned if c <= p)
            if nown:
                if second and nown > nprev:
                    # the pair's second panel is this rank's: its identity columns join the right-hand sides now, and the
                    # first panel's W is structurally zero there (the k-depth-1024 update reads those columns of it)
                    Wt2[:PW, nprev * PW:nown * PW].zero_()
                eng.solve_update2(wide, p, X, Wt2, None, nown * PANEL, second)
                w = L.width(p)
                X[p * PW:p * PW + w, :nown * PW].copy_(Wt2[second * PW:second * PW + w, :nown * PW])
            nprev = nown
        return X

    def kinv(self, Xl, out=None):
        """K^-1 = X^T X (lower tiles) for the owned block columns, np x 512 * owned panels: every owner broadcasts
        its block columns of X in turn (the next one in flight while the current one is consumed) and each rank forms
        the rows of that panel against its own columns on the MFMA tile engine (gpimhip_dist_kinv_update).
        out: where to (default: a new matrix); the training loop passes ``self.local`` -- the factor is dead by then.
        Only the LOWER tiles of the result are written: with ``out=self.local`` the tiles above the diagonal keep
        factor / K values, and this object no longer holds a factor (solve / logdet / inverse / gather_lower raise until
        the next factor()).  Xl (the workspace ``inverse`` returned) is valid until the next ``inverse`` call."""
        L, eng = self.layout, self.engine
        Kl = out if out is not None else self._matrix(L.np, L.local_cols)
        if out is not None and out.data_ptr() == self.local.data_ptr():
            self._factored = False

        def fill_of(c):
            def fill(buf):
                l0, r0 = L.local_col0(c), c * PW
                buf[r0:L.np, :L.width(c)].copy_(Xl[r0:, l0:l0 + L.width(c)])
            return fill
        # Several panels per launch (round 6): the first panels have few tiles -- 16 c of them for panel c on one GPU, an
        # eighth of that on eight -- each as deep as the whole matrix, and one panel at a time they leave most of the chip idle
        # (the pass ran at 57 TFLOP/s at N = 65536 where the single-GPU K^-1 launch reaches 68)
        G = KINV_GROUP if hasattr(eng, "kinv_update") and getattr(eng, "grouped_kinv", False) else 1
        if G > 1:
            for c0, wide, cnt in self._stream_groups(fill_of, G):
                eng.kinv_update(wide, c0, Xl, Kl, cnt)
        else:
            for c, buf in self._stream(fill_of):
                eng.kinv_update(buf, c, Xl, Kl)
        return Kl

    def nll(self, y):
        """1/2 y^T K^-1 y + 1/2 log det K + N/2 log 2 pi (no prior constant)."""
        alpha = self.solve(y)
        yv = torch.as_tensor(y, dtype=torch.float64).to(alpha.device)
        return 0.5 * float((yv * alpha).sum().item()) + 0.5 * self.logdet() + 0.5 * self.layout.n * math.log(2 * math.pi)

    def gather_lower(self):
        """The whole factor on every rank (tests and small problems only)."""
        self._need_factor("gather_lower")
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
        wpad = max(NB, (s1 - s0 + NB - 1) // NB * NB)      # the tile engine works on multiples of 128 columns
        Ks = torch.zeros((npd, wpad), dtype=torch.float64, device=dev)
        if s1 > s0:
            kmat(Xt[s0:s1].to(dev).contiguous(), Ks)
            mu = torch.empty((wpad,), dtype=torch.float64, device=dev)
            chol.engine.matvec_t(Ks[:N], alpha, mu)
            part[0, s0 - lo:s1 - lo] = mu[:s1 - s0]
        if with_sd:
            q = chol.solve_colsumsq(Ks)[:s1 - s0]
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


def exact_gp_fit(X, y, kernel="RBF", lengthscale=None, learning_rate=5e-2, iterations=100, seed=0, jitter=1e-5,
                 amplitude=None, group=None, verbose=0, u0=None):
    """Trains ONE exact GP on all N points across the ranks of the process group: the training loop of
    ``reconstructor.train`` (gpim/gpreg/gpr.py:170-217 -- MAP hyper-parameters under Uniform priors, Adam with the
    reference's learning rate) for covariances too large for one device.  X (N, d), y (N,); ``lengthscale`` = the
    prior bounds [[lo...], [hi...]] (or two scalars: isotropic) as in ``reconstructor``; every rank passes the same
    arguments and ends with the same hyper-parameters (the initial draw comes from a private CPU generator seeded with
    ``seed``; the gradient sums are all-reduced, everything after them is replicated arithmetic).

    Per Adam iteration: every rank builds its own column panels of K(u) (gpimhip_dist_kmat_cols), the block-column-
    cyclic factorisation (DistributedCholesky.factor), log det and alpha = K^-1 y (two distributed triangular
    solves), X = L^-1 (``inverse``), K^-1 = X^T X for the owned columns (``kinv``), the rank's share of
    sum_ij (K^-1 - alpha alpha^T)_ij dK_ij/dtheta (gpimhip_dist_grad_sums), ONE all-reduce of 8 doubles, and the
    chain rule + Adam step of the single-GPU path (gpimhip_dist_finalize).

    Returns (hyperparams, u): the dictionary ``reconstructor.hyperparams`` holds ("lengthscale", "variance", "noise"
    histories as numpy arrays, plus "loss") and the final unconstrained parameter vector (device tensor) -- feed the last
    history row to ``exact_gp_posterior``."""
    from . import _lib
    from .kernels import KernelSpec
    rank, world = _world()
    X = torch.as_tensor(X, dtype=torch.float64)
    N, d = X.shape
    if lengthscale is None:
        lengthscale = [[0.0] * d, [float(X.max()) / 2] * d]
    spec = KernelSpec(kernel, d, lengthscale, amplitude=amplitude, jitter=jitter)
    m = spec.struct()
    P = spec.n_params
    chol = DistributedCholesky(N, group=group)
    L, H = chol.layout, chol.engine.H
    dev = H.device
    lib = H.lib
    Xd = X.to(dev).contiguous()
    yd = torch.as_tensor(y, dtype=torch.float64).to(dev).contiguous()
    u = (spec.draw_initial_u(torch.Generator().manual_seed(seed)) if u0 is None
         else torch.as_tensor(u0, dtype=torch.float64).clone()).to(dev).contiguous()
    T = int(iterations)
    hist = torch.zeros((max(T, 1), P), dtype=torch.float64, device=dev)
    loss = torch.zeros((max(T, 1),), dtype=torch.float64, device=dev)
    # red[0..7]: the gradient sums, red[8]: this rank's sum log L_ii, red[9]: 1 if this rank met a non-positive pivot --
    # ONE all-reduce per iteration; quad = y^T alpha is replicated arithmetic.  Nothing is read back inside an iteration:
    # the loss and the failure count of iteration t reach the host through one pinned copy, awaited after iteration
    # t has been enqueued (three .item() synchronisations per iteration until round 4).
    red = torch.zeros((10,), dtype=torch.float64, device=dev)
    quad = torch.zeros((1,), dtype=torch.float64, device=dev)
    alpha_pad = torch.zeros((L.np,), dtype=torch.float64, device=dev)
    back = torch.zeros((2,), dtype=torch.float64).pin_memory() if dev.type == "cuda" else torch.zeros((2,), dtype=torch.float64)
    ld = chol.local.stride(0)
    eng = chol.engine
    for t in range(1, T + 1):
        for p in L.owned:
            l0 = L.local_col0(p)
            _lib.check(lib.gpimhip_dist_kmat_cols(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), p * PW, L.width(p),
                                                  ctypes.c_void_p(chol.local.data_ptr() + 8 * l0), ld))
        chol.factor(check=False)
        alpha = chol.solve(yd)
        torch.sum(yd * alpha, dim=0, keepdim=True, out=quad)
        alpha_pad[:N] = alpha
        Xl = chol.inverse()
        Kl = chol.kinv(Xl, out=chol.local)                    # (the factor is dead: K^-1 takes its place)
        _lib.check(lib.gpimhip_dist_grad_sums(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(u), _lib.ptr(Kl),
                                              Kl.stride(0), _lib.ptr(alpha_pad), _lib.ptr(red)))
        red[8:9].copy_(eng.half_logdet_owned())
        red[9:10].copy_((eng.info[0:1] != 0).to(torch.float64))
        if world > 1:
            dist.all_reduce(red, group=group)
        _lib.check(lib.gpimhip_dist_finalize_dev(H.h, ctypes.byref(m), N, _lib.ptr(u), _lib.ptr(red), _lib.ptr(quad),
                                                 float(learning_rate), t, ctypes.c_void_p(loss[t - 1:].data_ptr()), None,
                                                 ctypes.c_void_p(hist[t - 1].data_ptr())))
        back[0:1].copy_(red[9:10], non_blocking=True)
        back[1:2].copy_(loss[t - 1:t], non_blocking=True)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        if back[0].item() != 0:
            bad = torch.tensor([eng.failed_column()], dtype=torch.int64, device=dev)
            if world > 1:
                dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
            raise torch.linalg.LinAlgError("linalg.cholesky: the input is not positive-definite "
                                           "(leading minor of order %d)" % int(bad.item()))
        if verbose and rank == 0 and (t == 1 or t % 10 == 0 or t == T):
            print("iter: {} ... loss: {:.4f}".format(t - 1, float(back[1].item())))
    hcpu = hist[:T].cpu().numpy()
    hyper = {"variance": hcpu[:, 0], "lengthscale": hcpu[:, 1:1 + spec.n_ls], "noise": hcpu[:, 1 + spec.n_ls],
             "loss": loss[:T].cpu().numpy()}
    return hyper, u


def exact_gp_nll_grad(X, y, u, kernel="RBF", lengthscale=None, jitter=1e-5, amplitude=None, group=None):
    """Loss and d loss / du of the distributed model at the unconstrained parameters u (no Adam step): what
    gpimhip_nll_grad returns on one GPU.  Returns (loss, grad) as a float and a numpy vector, the same on every rank."""
    from . import _lib
    from .kernels import KernelSpec
    rank, world = _world()
    X = torch.as_tensor(X, dtype=torch.float64)
    N, d = X.shape
    spec = KernelSpec(kernel, d, lengthscale, amplitude=amplitude, jitter=jitter)
    m = spec.struct()
    chol = DistributedCholesky(N, group=group)
    L, H = chol.layout, chol.engine.H
    dev, lib = H.device, H.lib
    Xd = X.to(dev).contiguous()
    yd = torch.as_tensor(y, dtype=torch.float64).to(dev).contiguous()
    ud = torch.as_tensor(u, dtype=torch.float64).clone().to(dev).contiguous()
    ld = chol.local.stride(0)
    for p in L.owned:
        _lib.check(lib.gpimhip_dist_kmat_cols(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(ud), p * PW, L.width(p),
                                              ctypes.c_void_p(chol.local.data_ptr() + 8 * L.local_col0(p)), ld))
    chol.factor()
    half_logdet = 0.5 * chol.logdet()
    alpha = chol.solve(yd)
    quad = float((yd * alpha).sum().item())
    alpha_pad = torch.zeros((L.np,), dtype=torch.float64, device=dev)
    alpha_pad[:N] = alpha
    Kl = chol.kinv(chol.inverse())
    S = torch.zeros((8,), dtype=torch.float64, device=dev)
    _lib.check(lib.gpimhip_dist_grad_sums(H.h, ctypes.byref(m), _lib.ptr(Xd), N, _lib.ptr(ud), _lib.ptr(Kl), Kl.stride(0),
                                          _lib.ptr(alpha_pad), _lib.ptr(S)))
    if world > 1:
        dist.all_reduce(S, group=group)
    out = torch.zeros((1 + spec.n_params,), dtype=torch.float64, device=dev)
    _lib.check(lib.gpimhip_dist_finalize(H.h, ctypes.byref(m), N, _lib.ptr(ud), _lib.ptr(S), quad, half_logdet, 0.0, 0,
                                         ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8), None))
    o = out.cpu().numpy()
    return float(o[0]), o[1:]
