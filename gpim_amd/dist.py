"""
dist.py -- multi-GPU sharding of independent GP work units (one process per GPU, RCCL over xGMI).

The reference is single-process (no distributed backend).  The exact-GP hot path
shards in two embarrassingly parallel ways:
  * independent units -- spectral slices of a cube, frames of an image stack -- each a complete
    ``reconstructor(...).run()``; units are dealt round-robin to ranks, there is no collective on
    the data path, and one ``gather`` at the end brings (mean, sd) to rank 0;
  * acquisition candidates -- every rank sweeps a contiguous block of the test grid and an
    ``all_gather`` of (value, flat index) pairs yields the global arg-max / top-k on every rank.
Messages are a few KB to a few MB, so the collectives are latency-bound; nothing here depends
on xGMI link bandwidth.  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  Single-process runs need no initialisation.
    backend: "nccl" (= RCCL; the default on a GPU box), "gloo", or the environment variable GPIM_DIST_BACKEND.
    With gloo several ranks may share one GPU (local_rank is taken modulo the device count) -- RCCL refuses
    that -- which is how the multi-rank code paths are exercised on a one-GPU box (tests/test_gpu_dist2.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        local_rank %= max(1, torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("GPIM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _host_staged(t):
    """gloo moves device tensors only for broadcast and all_reduce: the other collectives stage through the host."""
    return t.is_cuda and dist.get_backend() == "gloo"


def barrier():
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def all_gather(parts, t, group=None):
    """dist.all_gather(parts, t) for either backend (parts: list of tensors like t, filled in place)."""
    if _host_staged(t):
        host = [torch.empty(p.shape, dtype=p.dtype) for p in parts]
        dist.all_gather(host, t.cpu(), group=group)
        for p, hp in zip(parts, host):
            p.copy_(hp)
    else:
        dist.all_gather(parts, t, group=group)


def gather(t, parts, dst=0, group=None):
    """dist.gather(t, gather_list=parts, dst) for either backend (parts: list on the root, None elsewhere)."""
    if _host_staged(t):
        host = [torch.empty(p.shape, dtype=p.dtype) for p in parts] if parts is not None else None
        dist.gather(t.cpu(), gather_list=host, dst=dst, group=group)
        if parts is not None:
            for p, hp in zip(parts, host):
                p.copy_(hp)
    else:
        dist.gather(t, gather_list=parts, dst=dst, group=group)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_units(n_units, rank=None, world_size=None):
    """Indices of the work units owned by `rank` (round-robin, so ragged costs average out)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    return list(range(rank, n_units, world_size))


def gather_to_root(local, n_units, unit_shape, device=None, dtype=torch.float64):
    """local: {unit index: tensor of shape unit_shape}.  Returns on rank 0 a tensor
    (n_units, *unit_shape) with every unit in place (None on other ranks).  One collective of a
    fixed-size slab per rank; ranks owning fewer units pad with NaN."""
    rank, ws = world()
    per = (n_units + ws - 1) // ws
    if device is None:
        device = next(iter(local.values())).device if local else torch.device("cpu")
    slab = torch.full((per,) + tuple(unit_shape), float("nan"), dtype=dtype, device=device)
    for slot, idx in enumerate(shard_units(n_units, rank, ws)):
        slab[slot] = local[idx].to(device=device, dtype=dtype)
    if ws == 1:
        parts = [slab]
    else:
        # one gather to rank 0 (RCCL ncclGather = grouped send/recv): only the root receives the slabs
        parts = [torch.empty_like(slab) for _ in range(ws)] if rank == 0 else None
        try:
            gather(slab, parts, dst=0)
        except (RuntimeError, NotImplementedError):      # a backend without gather: every rank takes every slab
            parts = [torch.empty_like(slab) for _ in range(ws)]
            all_gather(parts, slab)
        if rank != 0:
            return None
    full = torch.empty((n_units,) + tuple(unit_shape), dtype=dtype, device=device)
    for r in range(ws):
        for slot, idx in enumerate(shard_units(n_units, r, ws)):
            full[idx] = parts[r][slot]
    return full


def run_units(units, unit_fn, out_shape, device=None, concurrency=1):
    """Deal `units` over the ranks, run ``unit_fn(unit) -> (mean, sd)`` on the owned ones and
    gather; returns (mean_all, sd_all) of shape (len(units), *out_shape) on rank 0, else None.
    concurrency > 1 runs that many owned units at a time on host threads (the C ABI releases the
    GIL; each worker should use its own HIP stream) so that small problems overlap on one GPU."""
    rank, ws = world()
    owned = shard_units(len(units), rank, ws)

    def one(idx):
        mean, sd = unit_fn(units[idx])
        mean, sd = torch.as_tensor(mean), torch.as_tensor(sd)
        return idx, torch.stack([mean.reshape(out_shape), sd.reshape(out_shape)])

    if concurrency > 1 and len(owned) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=concurrency) as pool:
            mine = dict(pool.map(one, owned))
    else:
        mine = dict(one(i) for i in owned)
    if device is None and mine:
        device = next(iter(mine.values())).device
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    full = gather_to_root(mine, len(units), (2,) + tuple(out_shape), device=device)
    if full is None:
        return None
    return full[:, 0], full[:, 1]


def reconstruct_slices(cube, axis=-1, batch=16, return_hyperparams=False, handle=None, sparse_concurrency=4,
                       batch_concurrency=1, sparse_batch="auto", **recon_kwargs):
    """Independent GP reconstruction of every slice of a 3D / 4D cube along `axis` (configs C3 and
    C5 of SURVEY 8(d)).  Slices are sharded over the ranks; on each GPU the owned slices with the same
    number of observations advance in lock-step, `batch` at a time, through the batched engine
    (gpim_amd.batch) -- a single ~1000-point fit is latency-bound and leaves most of the chip idle.
    Per-slice results are those of ``reconstructor(X_slice, R_slice, X_full, **recon_kwargs).run()``.
    handle: an existing ``_lib.Handle`` for the batched fits (its workspace is reused between calls).
    sparse_batch: sparse (inducing-point) slices with the same number of observations advance in lock-step too, up to
        this many at a time (gpim_amd.batch.fit_predict_batch_sparse); 0 / 1: one reconstructor per slice; "auto": half
        of the owned slices (at most 8), so that two batches overlap (config C5 on one MI355X, tools/r5_c5_conc.py: one
        reconstructor per slice on four threads 0.54 s, one batch of 5 0.48, batches of 3 + 2 on two threads 0.40).
    sparse_concurrency: how many sparse batches (or single sparse slices) are fitted at the same time on one GPU (host
        threads, each with its own stream and handle).
    batch_concurrency: how many lock-step batches of exact GPs run at the same time on one GPU (own streams).
    batch="auto" picks both from the number of slices this rank owns.
    Returns (mean, sd) cubes on rank 0 (None elsewhere)."""
    from . import gprutils
    from .batch import fit_predict_batch
    from .gpr import reconstructor
    cube = np.moveaxis(np.asarray(cube), axis, 0)
    nunits = cube.shape[0]
    rank, ws = world()
    dev = handle.device if handle is not None else torch.device("cuda", torch.cuda.current_device())
    owned = shard_units(nunits, rank, ws)
    Xf = gprutils.get_full_grid(cube[0])
    mine, hyper = {}, {}
    recon_kwargs = dict(recon_kwargs)
    recon_kwargs.pop("verbose", None)

    def grid_of(R):
        # fully observed slices have no NaN pattern to encode (get_sparse_grid refuses them)
        return gprutils.get_sparse_grid(R) if np.isnan(R).any() else gprutils.get_full_grid(R)

    if recon_kwargs.get("sparse"):
        # Inducing-point models (config C5: slices of a 4D cube): one reconstructor per slice.  A sparse fit is a
        # stream of ~90 small launches per Adam iteration (1.6 ms at N = 6400, 534 inducing inputs: latency, not
        # throughput), so several slices run CONCURRENTLY, each on its own host thread with its own HIP stream and
        # library handle (the C ABI call that runs the whole training loop releases the GIL); every slice's
        # arithmetic is that of its stand-alone reconstructor.
        # Slices of equal size go through the lock-step batch of the sparse model (every launch carries all of them: the two
        # five-block factorisation chains per iteration are shared, the skinny products become chip-sized launches).
        kw_b = {k: v for k, v in recon_kwargs.items() if k not in ("sparse", "use_gpu")}
        if sparse_batch == "auto":
            sparse_batch = len(owned) if len(owned) < 3 else min(8, (len(owned) + 1) // 2)
        if sparse_batch and int(sparse_batch) > 1 and kw_b.get("precision", "double") == "double":
            from .batch import fit_predict_batch_sparse
            by_size = {}
            for i in owned:
                by_size.setdefault(int(np.count_nonzero(~np.isnan(cube[i]))), []).append(i)
            rest, sgroups = [], []
            for n_obs, idxs in sorted(by_size.items()):
                for s0 in range(0, len(idxs), int(sparse_batch)):
                    grp = idxs[s0:s0 + int(sparse_batch)]
                    if len(grp) < 2:
                        rest.extend(grp)
                    else:
                        sgroups.append(grp)

            def one_sparse_group(grp, own_stream):
                args = ([grid_of(cube[i]) for i in grp], [cube[i] for i in grp], Xf)
                try:
                    if own_stream:
                        with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                            out = fit_predict_batch_sparse(*args, **kw_b)      # (its own handle, closed on return)
                            torch.cuda.current_stream().synchronize()
                    else:
                        out = fit_predict_batch_sparse(*args, handle=handle, **kw_b)
                except _lib.NotPositiveDefiniteError:
                    # one slice's factorisation failed and froze the whole lock-step batch: the slices of this group go
                    # through one reconstructor each, so that only the failing slice raises (as in the reference's loop)
                    return (grp, None)
                return (grp,) + tuple(out)

            # several batches at a time (own host thread / stream / handle each): the two factorisation chains per
            # iteration of one batch run beside the products of another
            if int(sparse_concurrency) > 1 and len(sgroups) > 1 and handle is None:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=min(int(sparse_concurrency), len(sgroups))) as pool:
                    sdone = list(pool.map(lambda g: one_sparse_group(g, True), sgroups))
            else:
                sdone = [one_sparse_group(g, False) for g in sgroups]
            for res in sdone:
                if res[1] is None:
                    rest.extend(res[0])
                    continue
                grp, mean, sd, hist, hist_xu = res
                hist_h, xu_h = hist.cpu().numpy(), hist_xu.cpu().numpy()
                n_ls = hist_h.shape[2] - 2 - (1 if str(kw_b.get("kernel", "RBF")) == "RationalQuadratic" else 0)
                iso = bool(kw_b.get("isotropic"))
                for k, i in enumerate(grp):
                    mine[i] = torch.stack([mean[k], sd[k]])
                    hyper[i] = {"lengthscale": [float(r[1]) if iso else r[1:1 + n_ls].tolist() for r in hist_h[k]],
                                "noise": [float(r[1 + n_ls]) for r in hist_h[k]],
                                "variance": [float(r[0]) for r in hist_h[k]],
                                "inducing_points": list(xu_h[k])}
            owned = rest

        def one_sparse(i):
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                rec = reconstructor(grid_of(cube[i]), cube[i], Xf, verbose=0, **recon_kwargs)
                rec.train()
                rec.predict()
                out = torch.stack(list(rec._last_pred)).reshape((2,) + tuple(cube.shape[1:]))
                torch.cuda.current_stream().synchronize()
                return i, out, rec.hyperparams
        nthreads = max(1, min(int(sparse_concurrency), len(owned))) if owned else 0
        if nthreads == 0:
            results = []
        elif nthreads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=nthreads) as pool:
                results = list(pool.map(one_sparse, owned))
        else:
            results = [one_sparse(i) for i in owned]
        for i, out, hp in results:
            mine[i], hyper[i] = out, hp
        owned = []
    by_n = {}
    for i in owned:
        by_n.setdefault(int(np.count_nonzero(~np.isnan(cube[i]))), []).append(i)
    if batch == "auto":
        # measured on one MI355X (N = 1207 per slice, tools/r3_c3c.py): 8 slices 4 x 2 concurrent 0.27 s (one batch of
        # 8: 0.32), 16 slices 4 x 4 0.35 (0.45), 32 slices 16 x 2 0.64 (0.68), 64 slices 16 x 4 0.99 (1.11)
        # (round 6, tools/r6_c3_queues.py: 8 slices 2 x 4 0.209 s, 4 x 2 0.216-0.226, 1 x 8 0.34; more than four concurrent
        # batches are slower whatever GPU_MAX_HW_QUEUES says: 64 slices 8 x 8 1.12, 4 x 16 1.19 against 16 x 4 0.889)
        batch = 2 if len(owned) <= 8 else (4 if len(owned) <= 16 else 16)
        batch_concurrency = min(4, max(1, (len(owned) + batch - 1) // batch))
    groups = []
    for n_obs, idxs in sorted(by_n.items()):
        for s in range(0, len(idxs), batch):
            groups.append(idxs[s:s + batch])

    def one_group(grp, own_stream):
        Xs = [grid_of(cube[i]) for i in grp]
        if own_stream:
            with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(device=dev)):
                mean, sd, hist = fit_predict_batch(Xs, [cube[i] for i in grp], Xf, **recon_kwargs)
                torch.cuda.current_stream().synchronize()
        else:
            mean, sd, hist = fit_predict_batch(Xs, [cube[i] for i in grp], Xf, handle=handle, **recon_kwargs)
        return grp, mean, sd, hist

    # several lock-step batches at a time (each on its own host thread / HIP stream / handle): the latency-bound
    # launches of one batch (diagonal-block factorisations, panel solves) overlap the tile products of another
    if batch_concurrency > 1 and len(groups) > 1 and handle is None:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(int(batch_concurrency), len(groups))) as pool:
            done = list(pool.map(lambda g: one_group(g, True), groups))
    else:
        done = [one_group(g, False) for g in groups]
    for grp, mean, sd, hist in done:
        for k, i in enumerate(grp):
            mine[i] = torch.stack([mean[k], sd[k]])
            hyper[i] = hist[k].cpu().numpy()
    full = gather_to_root(mine, nunits, (2,) + tuple(cube.shape[1:]), device=dev)
    if full is None:
        return None
    out = (np.moveaxis(full[:, 0].cpu().numpy(), 0, axis), np.moveaxis(full[:, 1].cpu().numpy(), 0, axis))
    if return_hyperparams:
        return out + ([hyper.get(i) for i in range(nunits)],)
    return out


def candidate_block(M, rank=None, world_size=None):
    """Contiguous [start, stop) block of the M test-grid candidates owned by `rank`."""
    if rank is None or world_size is None:
        rank, world_size = world()
    per = (M + world_size - 1) // world_size
    return min(rank * per, M), min((rank + 1) * per, M)


def global_topk(local_vals, local_idx, k, nan_first=False):
    """local_vals / local_idx: this rank's descending top-k (values, GLOBAL flat indices), padded
    with -inf / -1 to length k.  All-gathers the 2*k*world numbers and returns the global
    descending top-k on every rank (ties: larger flat index first, as gpimhip_topk; NaN values rank
    above everything when nan_first, like the un-masked ranking of boptim.py:303-306)."""
    rank, ws = world()
    vals = torch.as_tensor(local_vals, dtype=torch.float64).reshape(-1)
    idx = torch.as_tensor(local_idx, dtype=torch.int64).reshape(-1).to(vals.device)
    if ws > 1:
        vs = [torch.empty_like(vals) for _ in range(ws)]
        ix = [torch.empty_like(idx) for _ in range(ws)]
        all_gather(vs, vals)
        all_gather(ix, idx)
        vals, idx = torch.cat(vs), torch.cat(ix)
    keep = idx >= 0
    if not nan_first:
        keep = keep & ~torch.isnan(vals)
    vals, idx = vals[keep], idx[keep]
    # descending by (NaN first, value, flat index): three stable sorts from the least significant key up,
    # on the device the pairs live on
    isnan = torch.isnan(vals)
    order = torch.sort(idx, descending=True, stable=True).indices
    order = order[torch.sort(torch.where(isnan, torch.full_like(vals, float("inf")), vals)[order], descending=True,
                             stable=True).indices]
    order = order[torch.sort(isnan[order].to(torch.int8), descending=True, stable=True).indices][:k]
    return vals[order], idx[order]


def all_gather_blocks(block, M):
    """block: this rank's slice (candidate_block) of a length-M vector; returns the full vector on
    every rank."""
    rank, ws = world()
    if ws == 1:
        return block
    per = (M + ws - 1) // ws
    pad = torch.full((per,), float("nan"), dtype=block.dtype, device=block.device)
    pad[:block.numel()] = block
    parts = [torch.empty_like(pad) for _ in range(ws)]
    all_gather(parts, pad)
    return torch.cat(parts)[:M] if per * ws == M else torch.cat(
        [parts[r][:max(0, min(per, M - r * per))] for r in range(ws)])
