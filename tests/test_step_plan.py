"""The launch plan of the blocked Cholesky with the triangular inverse riding in its launches (cholstep.hip:
plan_updates / plan_inverse; replaces torch.linalg.cholesky + the solves at gpim/gpreg/gpr.py:192-193,248) replayed on
the HOST with small blocks: the plan is expressed in block indices, so an interpreter with 2x2 or 3x3 blocks executes
exactly the dependency structure the GPU launches have.  Launch semantics: every tile operation of a launch reads the
state the launch started from (hosted workgroups run concurrently with each other and with the factorisation of the
diagonal block), so a schedule that hands an operation to a launch too early, lets two operations of one launch write
the same tile, or reads a tile another operation of the same launch writes, fails here.  No GPU involved."""
import ctypes

import numpy as np
import pytest


def plan(nb, with_inverse):
    from gpim_amd import _lib
    lib = _lib.load()
    n = ctypes.c_int64()
    assert lib.gpimhip_step_plan_host(nb, with_inverse, None, 0, ctypes.byref(n)) == 0
    buf = np.zeros((max(n.value, 1), 6), dtype=np.int32)
    assert lib.gpimhip_step_plan_host(nb, with_inverse, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n.value,
                                      ctypes.byref(n)) == 0
    return buf[:n.value]


def replay(nb, bs, with_inverse, seed=0):
    rng = np.random.default_rng(seed)
    n = nb * bs
    G = rng.standard_normal((n, n))
    K = G @ G.T / n + 2.0 * np.eye(n)
    A = np.tril(K).copy()
    A += np.tril(K, -1).T * 0.0          # (upper part unused)
    Tm = np.full((n, n), np.nan)
    rec = plan(nb, with_inverse)
    by_launch = {}
    for r in rec:
        by_launch.setdefault(int(r[0]), []).append(tuple(int(v) for v in r[1:]))
    blk = lambda M, i, j: M[i * bs:(i + 1) * bs, j * bs:(j + 1) * bs]
    W = 4

    def hosted(ops, A, Tm):
        A0, T0 = A.copy(), Tm.copy()
        written = set()
        reads = set()
        for ci, cj, k0, k1, kind in ops:
            assert 0 <= k0 < k1 and cj <= ci
            dst = ("T" if kind in (1, 2) else "A", ci, cj)
            assert dst not in written, "two operations of one launch write %s" % (dst,)
            written.add(dst)
            acc = np.zeros((bs, bs))
            for k in range(k0, k1):
                if kind == 0:
                    acc += blk(A0, ci, k) @ blk(A0, cj, k).T
                    reads.update({("A", ci, k), ("A", cj, k)})
                elif kind in (1, 2):
                    acc += blk(A0, ci, k) @ blk(A0, k, cj)
                    reads.update({("A", ci, k), ("A", k, cj)})
                else:
                    acc += blk(A0, ci, k) @ blk(T0, k, cj)
                    reads.update({("A", ci, k), ("T", k, cj)})
            if kind == 0:
                blk(A, ci, cj)[...] = blk(A0, ci, cj) - acc
            elif kind == 1:
                blk(Tm, ci, cj)[...] = acc
            elif kind == 2:
                blk(Tm, ci, cj)[...] = blk(T0, ci, cj) + acc
            elif kind == 3:
                blk(A, ci, cj)[...] = -acc
            else:
                blk(A, ci, cj)[...] = blk(A0, ci, cj) - acc
        assert not (written & reads), "an operation reads what another operation of the same launch writes"
        return written

    Lref = np.linalg.cholesky(K)
    for j in range(nb):
        ops = by_launch.get(j, [])
        for o in ops:
            assert o[3] <= j or o[4] != 0, "a trailing update uses a block column that is not final"
        written = hosted(ops, A, Tm)
        assert ("A", j, j) not in written
        # factorisation role of the same launch (concurrent with the hosted tiles: it only touches block (j, j))
        d = np.tril(blk(A, j, j))
        d = np.tril(d) + np.tril(d, -1).T
        Lj = np.linalg.cholesky(d)
        Dinv = np.linalg.inv(Lj)
        blk(A, j, j)[...] = Dinv if with_inverse else Lj
        # panel solve, then the next diagonal tiles
        for i in range(j + 1, nb):
            blk(A, i, j)[...] = blk(A, i, j) @ Dinv.T
        # D_j: the NEXT diagonal tile only (its older window columns are hosted tile operations of launch j)
        if j + 1 < nb:
            blk(A, j + 1, j + 1)[...] -= blk(A, j + 1, j) @ blk(A, j + 1, j).T
    for l in sorted(k for k in by_launch if k >= nb):
        hosted(by_launch[l], A, Tm)
    want = np.linalg.inv(Lref) if with_inverse else Lref
    got = np.tril(A)
    if not with_inverse:
        # diagonal blocks hold the factor's block (lower)
        pass
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 1e-10 * scale, np.abs(got - want).max()
    return rec


@pytest.mark.parametrize("nb", [1, 2, 3, 5, 8, 10, 13, 33, 40])
def test_plan_with_inverse_small(ensure_built, nb):
    replay(nb, 2, 1, seed=nb)


@pytest.mark.parametrize("nb", [5, 33])
def test_plan_factor_only(ensure_built, nb):
    replay(nb, 2, 0, seed=nb)


@pytest.mark.parametrize("nb,with_inverse", [(67, 1), (96, 1), (128, 0), (130, 1)])
def test_plan_large(ensure_built, nb, with_inverse):
    """nb >= 64: the full-round hosting policy with per-tile pending ranges (and the inverse in the chain-bound tail)."""
    rec = replay(nb, 1, with_inverse, seed=nb)
    upd = rec[rec[:, 5] == 0]
    if nb >= 96:
        assert (upd[:, 4] - upd[:, 3]).max() >= 8        # deferred tiles come back deeper
    if with_inverse:
        hosted_inv = rec[(rec[:, 5] > 0) & (rec[:, 0] < nb)]
        assert len(hosted_inv) > 0                        # part of the inverse rides in the step launches


@pytest.mark.parametrize("nb", [10, 33, 47, 128])
def test_lists_are_dispatched_deepest_first(ensure_built, nb):
    """Dispatch order = list order.  A launch of more workgroups than the chip has slots packs two shallow chunks into one
    slot only if the deep ones go first: the list of every launch (trailing updates and the inverse's chunks together)
    is sorted by k-depth, deepest first (N = 4212: launches of 260-290 quadrants 49 -> 38 us when this was fixed)."""
    rec = plan(nb, 1)
    for launch in np.unique(rec[:, 0]):
        r = rec[rec[:, 0] == launch]
        depth = r[:, 4] - r[:, 3]
        assert (np.diff(depth) <= 0).all(), int(launch)
