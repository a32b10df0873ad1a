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
