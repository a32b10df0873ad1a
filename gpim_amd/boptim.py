"""
boptim.py -- ``boptimizer``: GP-based Bayesian optimisation on a grid.

Host-side mirror of the reference's gpim/gpbayes/boptim.py:22-485 (SURVEY 8(a) rows a14-a15,
8(b)): same constructor signature and kwargs, same public attributes (``indices_all``,
``vals_all``, ``target_func_vals``, ``gp_predictions``, ``surrogate_model``) and methods.
The surrogate is a ``gpim_amd.reconstructor`` (HIP engine); the acquisition sweep and the
descending top-``batch_size`` ranking run on the GPU (gpimhip_acq / gpimhip_topk); the
revisit / distance-memory filter, batch thinning and bookkeeping are small host loops as in the
reference.  Ties in the ranking are resolved "larger flat index first" (the reference inherits
numpy's unspecified introsort order).
"""
import ctypes
import types

import numpy as np
import torch

from . import _lib, acqfunc, gprutils
from .gpr import reconstructor

_F64 = torch.float64


class _LazyMaps(list):
    """``boptimizer.gp_predictions``: one (mean, sd) pair of numpy maps per exploration step, like the
    reference's list -- but the maps of the built-in acquisition path stay on the GPU until somebody looks
    at them (indexing, iteration, saving), so an exploration step makes no device-to-host copy of a map."""

    class _Pending:
        def __init__(self, mean_d, sd_d, shape, np_dtype):
            self.mean_d, self.sd_d, self.shape, self.np_dtype = mean_d, sd_d, shape, np_dtype

        def get(self):
            both = torch.stack([self.mean_d, self.sd_d]).cpu().numpy().astype(self.np_dtype, copy=False)
            return both[0].reshape(self.shape), both[1].reshape(self.shape)

    def _mat(self, i):
        v = list.__getitem__(self, i)
        if isinstance(v, _LazyMaps._Pending):
            v = v.get()
            list.__setitem__(self, i, v)
        return v

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._mat(j) for j in range(*i.indices(len(self)))]
        return self._mat(i if i >= 0 else len(self) + i)

    def __iter__(self):
        return (self._mat(i) for i in range(len(self)))

    def __reversed__(self):
        return (self._mat(i) for i in range(len(self) - 1, -1, -1))

    def copy(self):
        return [self._mat(i) for i in range(len(self))]

    def __reduce__(self):                   # pickles (np.save of the results dict) as a plain list of arrays
        return (list, (list(iter(self)),))


class boptimizer:
    """
    Args mirror the reference: ``boptimizer(X_seed, y_seed, X_full, target_function,
    acquisition_function='cb', exploration_steps=10, batch_size=100, batch_update=False,
    kernel='RBF', lengthscale=None, sparse=False, indpoints=None, gp_iterations=1000, seed=0,
    **kwargs)`` with kwargs verbose, use_gpu (ignored), learning_rate, jitter (default 1e-6),
    isotropic, precision, alpha, beta, xi, dscale, batch_dscale, batch_out_max, gamma, memory,
    exit_strategy, mask, extent, simulate_measurement, y_true, save_checkpoints, filename.

    gpim_amd extension: ``shard_candidates=True`` (with torch.distributed initialised, one process
    per GPU) makes every rank sweep only its contiguous block of the test grid; an all-gather of
    each rank's top-``batch_size`` (value, index) pairs gives all ranks the same global ranking
    (SURVEY 8(e)).  Training stays replicated -- it is deterministic, so all ranks hold the same
    hyper-parameters -- and every rank evaluates the same next point.
    """

    def __init__(self, X_seed, y_seed, X_full, target_function, acquisition_function='cb',
                 exploration_steps=10, batch_size=100, batch_update=False, kernel='RBF',
                 lengthscale=None, sparse=False, indpoints=None, gp_iterations=1000, seed=0,
                 **kwargs):
        self.verbose = kwargs.get("verbose", 1)
        self.use_gpu = kwargs.get("use_gpu", False)
        learning_rate = kwargs.get("learning_rate", 5e-2)
        jitter = kwargs.get("jitter", 1.0e-6)
        isotropic = kwargs.get("isotropic", False)
        self.precision = kwargs.get("precision", "double")
        self.surrogate_model = reconstructor(
            X_seed, y_seed, X_full, kernel, lengthscale, sparse, indpoints,
            learning_rate, gp_iterations, self.use_gpu, self.verbose, seed,
            isotropic=isotropic, precision=self.precision, jitter=jitter)
        self.X_sparse = X_seed.copy()
        self.surrogate_model._trained_on = self.X_sparse        # same content as X_seed (see _observed_rows_d)
        self.y_sparse = y_seed.copy()
        self.X_full = X_full
        self.target_function = target_function
        self.acquisition_function = acquisition_function
        self.exploration_steps = exploration_steps
        self.batch_update = batch_update
        self.batch_size = batch_size
        self.simulate_measurement = kwargs.get("simulate_measurement", False)
        if self.simulate_measurement:
            self.y_true = kwargs.get("y_true")
            if self.y_true is None:
                raise AssertionError("To simulate measurements, add ground truth ('y_true)")
        self.extent = kwargs.get("extent", None)
        self.alpha, self.beta = kwargs.get("alpha", 0), kwargs.get("beta", 1)
        self.xi = kwargs.get("xi", 0.01)
        self.dscale = kwargs.get("dscale", None)
        self.batch_dscale = kwargs.get("batch_dscale", None)
        self.batch_out_max = kwargs.get("batch_out_max", 10)
        self.gamma = kwargs.get("gamma", 0.8)
        self.points_mem = kwargs.get("memory", 10)
        self.exit_strategy = kwargs.get("exit_strategy", 1)
        self.mask = kwargs.get("mask", None)
        self.save_checkpoints = kwargs.get("save_checkpoints", False)
        self.filename = kwargs.get("filename", "./boptim_results")
        self.shard_candidates = kwargs.get("shard_candidates", False)
        self.indices_all, self.vals_all = [], []
        self.target_func_vals, self.gp_predictions = [y_seed.copy()], _LazyMaps()
        self._mask_d = None
        self._Xfull_d = None
        self._map_slabs, self._maps_used = [], 0

    # ------------------------------------------------------------------ posterior update
    def update_posterior(self):
        """Swap the training set in place and train again (warm-started hyper-parameters,
        fresh Adam) -- boptim.py:239-251."""
        X_new, y_new = gprutils.prepare_training_data(self.X_sparse, self.y_sparse, precision=self.precision)
        self.surrogate_model.model.X = X_new
        self.surrogate_model.model.y = y_new
        self.surrogate_model.train(verbose=self.verbose)
        self.surrogate_model._trained_on = self.X_sparse

    def evaluate_function(self, indices, y_measured=None):
        """Evaluate the target at the new point(s) and refresh the sparse grid (boptim.py:253-276)."""
        indices = [indices] if not self.batch_update else indices
        for idx in indices:
            pos = tuple(idx)
            if self.simulate_measurement:
                self.y_sparse[pos] = self.y_true[pos]
            elif y_measured is not None:
                self.y_sparse[pos] = y_measured[pos]
            else:
                arg = pos if self.extent is None else tuple(i + e[0] for i, e in zip(idx, self.extent))
                self.y_sparse[pos] = self.target_function(arg)
        self.X_sparse = gprutils.get_sparse_grid(self.y_sparse, self.extent)
        self.target_func_vals.append(self.y_sparse.copy())

    # ------------------------------------------------------------------ ranking
    def _rank(self, acq):
        """Descending top-``batch_size`` of the acquisition map on the GPU.  Without a mask NaNs
        rank first (np.argsort puts them last, the reference then reverses); with a mask the
        product mask*acq is ranked and NaNs are dropped (boptim.py:303-315)."""
        sm = self.surrogate_model
        handle = sm._handle
        acq_d = getattr(sm, "_last_acq", None)
        if acq_d is None or acq_d.numel() != acq.size:
            acq_d = torch.as_tensor(np.ascontiguousarray(acq), dtype=_F64).reshape(-1).to(handle.device)
        sm._last_acq = None
        return self._rank_device(acq_d, acq.shape)

    def _retain_maps(self, mean_d, sd_d):
        """Copies the step's posterior maps into a slab allocated 16 steps at a time and returns views of it:
        the per-step tensors then go back to torch's caching allocator and are reused by the next step
        (holding on to them made every step pay fresh hipMallocs, ~10x the cost of the step itself)."""
        slot = self._maps_used % 16
        if slot == 0:
            # the previous slab is full: its 16 steps' maps go to the host now and the slab is released, so the
            # device holds at most one slab (16 x 2 x M values) however long the exploration runs
            for i in range(len(self.gp_predictions)):
                self.gp_predictions._mat(i)
            self._map_slabs.clear()
            self._map_slabs.append(torch.empty((16, 2, mean_d.numel()), dtype=mean_d.dtype, device=mean_d.device))
        slab = self._map_slabs[-1]
        if slab.shape[2] != mean_d.numel():                      # grid changed size: start a new slab
            for i in range(len(self.gp_predictions)):
                self.gp_predictions._mat(i)
            self._map_slabs.clear()
            self._map_slabs.append(torch.empty((16, 2, mean_d.numel()), dtype=mean_d.dtype, device=mean_d.device))
            slab, slot, self._maps_used = self._map_slabs[-1], 0, 0
        slab[slot, 0].copy_(mean_d)
        slab[slot, 1].copy_(sd_d)
        self._maps_used += 1
        return slab[slot, 0], slab[slot, 1]

    def _observed_rows_d(self):
        """Device rows of the observed points of ``X_sparse``: the surrogate's training inputs whenever it was last
        fitted to this very grid (update_posterior / the constructor do that); None otherwise."""
        sm = self.surrogate_model
        Xd = getattr(sm, "_Xd", None)
        if Xd is None or sm.do_sparse or sm.do_structured or getattr(sm, "_trained_on", None) is not self.X_sparse:
            return None
        return Xd

    def _device_mask(self, n):
        if self.mask is None:
            return None
        if self._mask_d is None or self._mask_d.numel() != n:
            dev = self.surrogate_model._handle.device
            self._mask_d = torch.as_tensor(np.ascontiguousarray(self.mask), dtype=_F64).reshape(-1).to(dev)
        return self._mask_d

    def _rank_device(self, acq_d, grid_shape, masked=False):
        """Top-``batch_size`` of a device-resident acquisition map (flattened); returns host lists.
        masked: the map has been multiplied by the mask already."""
        handle = self.surrogate_model._handle
        keep_nan = 1
        if self.mask is not None:
            if not masked:
                acq_d = self._device_mask(acq_d.numel()) * acq_d
            keep_nan = 0
        k = int(min(self.batch_size, acq_d.numel()))
        # one buffer, one device-to-host copy: [count | k indices | k values (as bits)]
        buf = torch.zeros((2 * k + 1,), dtype=torch.int64, device=handle.device)
        base = buf.data_ptr()
        _lib.check(handle.lib.gpimhip_topk(handle.h, _lib.ptr(acq_d.contiguous()), acq_d.numel(), k, keep_nan,
                                           ctypes.c_void_p(base + 8 * (k + 1)), ctypes.c_void_p(base + 8),
                                           ctypes.c_void_p(base)))
        host = buf.cpu().numpy()
        n = int(host[0])
        flat = host[1:1 + n]
        vals_list = host[k + 1:k + 1 + n].view(np.float64).tolist()
        indices_list = np.stack(np.unravel_index(flat, tuple(grid_shape)), axis=-1).tolist()
        return vals_list, indices_list

    def next_point(self):
        """Acquisition sweep + ranking (+ batch thinning) -- boptim.py:278-324."""
        if self.verbose:
            print("Computing acquisition function...")
        sm = self.surrogate_model
        sm._last_acq = None
        af = self.acquisition_function
        if self.shard_candidates and af in ('cb', 'ei', 'poi'):
            vals_list, indices_list = self._next_point_sharded()
            if not self.batch_update:
                return vals_list, indices_list
            radius = self.batch_dscale
            if radius is None:
                radius = sm.model.kernel.lengthscale.mean().item()
            return self.update_points(vals_list, indices_list, radius)
        if af in ('cb', 'ei', 'poi'):
            # built-in acquisition: prediction, incumbent, sweep, mask and ranking all stay on the GPU;
            # only the batch_size ranked (value, index) pairs come back
            p0, p1 = (self.alpha, self.beta) if af == 'cb' else (0.0, 0.0)
            if self._Xfull_d is None:
                self._Xfull_d = sm._to_device(gprutils.prepare_test_data(np.asarray(self.X_full), precision=self.precision))
            acq_d, mean_d, sd_d = acqfunc.acquisition_on_device(sm, af, self.X_full, self.X_sparse, p0, p1, self.xi,
                                                                Xf_d=self._Xfull_d,
                                                                mask_d=self._device_mask(self._Xfull_d.shape[0]),
                                                                Xobs_d=self._observed_rows_d())
            grid_shape = tuple(np.shape(self.X_full)[1:])
            mean_d, sd_d = self._retain_maps(mean_d, sd_d)
            self.gp_predictions.append(_LazyMaps._Pending(mean_d, sd_d, grid_shape, sm._np_out))
            vals_list, indices_list = self._rank_device(acq_d, grid_shape, masked=True)
        elif isinstance(af, types.FunctionType):
            acq, pred = af(sm, self.X_full, self.X_sparse)
            sm._last_acq = None
            self.gp_predictions.append(pred)
            vals_list, indices_list = self._rank(np.asarray(acq))
        else:
            raise NotImplementedError(
                "Choose between 'cb', 'ei', and 'poi' acquisition functions or define your own")
        if not self.batch_update:
            return vals_list, indices_list
        radius = self.batch_dscale
        if radius is None:
            radius = sm.model.kernel.lengthscale.mean().item()
        return self.update_points(vals_list, indices_list, radius)

    def _next_point_sharded(self):
        """Acquisition sweep over this rank's block of candidates + global top-k (RCCL all-gather)."""
        from . import dist as gdist
        sm = self.surrogate_model
        handle = sm._handle
        grid_shape = self.X_full.shape[1:]
        Xflat = np.asarray(self.X_full).reshape(self.X_full.shape[0], -1)
        M = Xflat.shape[1]
        lo, hi = gdist.candidate_block(M)
        Xblk = Xflat[:, lo:hi]
        af = self.acquisition_function
        if af == 'cb':
            acq, pred = acqfunc.confidence_bound(sm, Xblk, alpha=self.alpha, beta=self.beta)
        elif af == 'ei':
            acq, pred = acqfunc.expected_improvement(sm, Xblk, self.X_sparse, xi=self.xi)
        else:
            acq, pred = acqfunc.probability_of_improvement(sm, Xblk, self.X_sparse, xi=self.xi)
        acq_d = sm._last_acq
        sm._last_acq = None
        if acq_d is None:
            acq_d = torch.as_tensor(np.ascontiguousarray(acq), dtype=_F64).reshape(-1).to(handle.device)
        keep_nan = 1
        if self.mask is not None:
            mblk = np.asarray(self.mask).reshape(-1)[lo:hi]
            acq_d = torch.as_tensor(np.ascontiguousarray(mblk), dtype=_F64).to(handle.device) * acq_d
            keep_nan = 0
        k = int(min(self.batch_size, M))
        kl = int(min(k, max(hi - lo, 1)))
        vals = torch.full((k,), float("-inf"), dtype=_F64, device=handle.device)
        idx = torch.full((k,), -1, dtype=torch.int64, device=handle.device)
        if hi > lo:
            lv = torch.empty((kl,), dtype=_F64, device=handle.device)
            li = torch.empty((kl,), dtype=torch.int64, device=handle.device)
            cnt = torch.zeros((1,), dtype=torch.int64, device=handle.device)
            _lib.check(handle.lib.gpimhip_topk(handle.h, _lib.ptr(acq_d.contiguous()), acq_d.numel(), kl, keep_nan,
                                               _lib.ptr(lv), _lib.ptr(li), _lib.ptr(cnt)))
            n = int(cnt.item())
            vals[:n] = lv[:n]
            idx[:n] = li[:n] + lo
        gv, gi = gdist.global_topk(vals, idx, k, nan_first=bool(keep_nan))
        # the stored GP prediction is the full map on every rank
        mean_d, sd_d = sm._last_pred if sm._last_pred is not None else (None, None)
        mean_full = gdist.all_gather_blocks(torch.as_tensor(pred[0], dtype=_F64).reshape(-1).to(handle.device), M)
        sd_full = gdist.all_gather_blocks(torch.as_tensor(pred[1], dtype=_F64).reshape(-1).to(handle.device), M)
        self.gp_predictions.append((mean_full.cpu().numpy().reshape(grid_shape),
                                    sd_full.cpu().numpy().reshape(grid_shape)))
        flat = gi.cpu().numpy()
        return gv.cpu().numpy().tolist(), np.stack(np.unravel_index(flat, grid_shape), axis=-1).tolist()

    def update_points(self, acqfunc_values, indices, dscale):
        """Thin a ranked batch so that kept points are farther than ``dscale`` apart, pad with random
        members of the batch -- boptim.py:326-376.  The greedy maximum-and-suppress loop (the reference's
        cKDTree ball queries) runs on the GPU (gpimhip_thin_batch); the random padding draws from
        ``np.random`` on the host like the reference."""
        _, val = self.checkvalues(indices, acqfunc_values)
        first = np.where(np.array(acqfunc_values) == val)[0][0]
        vals = np.array(acqfunc_values, dtype=np.float64)[first:]
        pts = np.vstack(indices)[first:]
        n, d = pts.shape
        if n > 1024 or d > _lib.MAX_DIM:
            # beyond the device kernel's limits (gpimhip_thin_batch: <= 1024 candidates, d <= 4): the same greedy
            # maximum-and-suppress loop on the host
            kept_ids = self._thin_host(vals, pts, float(dscale), int(self.batch_out_max))
            return self._pad_batch(vals, pts, kept_ids)
        handle = self.surrogate_model._handle
        dev = handle.device
        shape = pts.max(axis=0).astype(np.int64) + 1                  # any box containing the candidates
        flat = np.ravel_multi_index(tuple(pts.T.astype(np.int64)), tuple(shape))
        vals_d = torch.from_numpy(vals).to(dev)
        flat_d = torch.from_numpy(flat.astype(np.int64)).to(dev)
        shape_d = torch.from_numpy(shape).to(dev)
        keep_d = torch.empty((self.batch_out_max,), dtype=torch.int32, device=dev)
        nkeep_d = torch.zeros((1,), dtype=torch.int32, device=dev)
        _lib.check(handle.lib.gpimhip_thin_batch(handle.h, _lib.ptr(vals_d), _lib.ptr(flat_d), int(n), int(d),
                                                 _lib.ptr(shape_d), float(dscale), int(self.batch_out_max),
                                                 _lib.ptr(keep_d), _lib.ptr(nkeep_d)))
        kept_ids = keep_d[:int(nkeep_d.item())].cpu().numpy().astype(np.int64)
        return self._pad_batch(vals, pts, kept_ids)

    @staticmethod
    def _thin_host(vals, pts, dscale, max_out):
        """boptim.py:352-365 of the reference without the k-d tree: keep the largest remaining value, drop every
        candidate within Euclidean distance dscale of it (inclusive, as cKDTree.query_ball_point), repeat."""
        alive = np.ones(len(vals), dtype=bool)
        work = vals.astype(np.float64).copy()
        P = pts.astype(np.float64)
        kept = []
        while alive.any() and len(kept) < max_out:
            i = int(np.argmax(np.where(alive, work, -np.inf)))
            kept.append(i)
            alive &= np.linalg.norm(P - P[i], axis=1) > dscale
        return np.asarray(kept, dtype=np.int64)

    def _pad_batch(self, vals, pts, kept_ids):
        kept_vals = vals[kept_ids].tolist()
        out = pts[kept_ids].tolist()
        if len(out) < self.batch_out_max:
            if self.verbose == 2:
                print("Adding {} random indices".format(self.batch_out_max - len(out)))
            rnd = np.random.randint(0, len(vals), self.batch_out_max - len(out))
            out.extend(pts[rnd].tolist())
            kept_vals.extend(vals[rnd].tolist())
        return kept_vals, out

    def checkvalues(self, idx_list, val_list):
        """First ranked point that was not queried before and is not within the decaying
        distance memory of the last ``memory`` queried points -- boptim.py:378-429."""
        dscale = 0 if self.dscale is None else self.dscale

        def in_memory(idx):
            recent = self.indices_all[-self.points_mem:]
            dists = [np.linalg.norm(np.array(idx) - np.array(p)) for p in recent][::-1]
            limits = [dscale * self.gamma ** i for i in range(len(recent))]
            return any(not (d > l) for d, l in zip(dists, limits))

        pos = 0
        if self.verbose == 2:
            print('Acquisition function max value {} at {}'.format(val_list[pos], idx_list[pos]))
        if len(self.indices_all) == 0:
            return idx_list[pos], val_list[pos]
        while any(p == idx_list[pos] for p in self.indices_all) or in_memory(idx_list[pos]):
            if self.verbose == 2:
                print("Finding the next max point...")
            pos += 1
            if pos == len(idx_list):
                pos = np.random.randint(0, len(idx_list)) if self.exit_strategy else -1
                if self.verbose == 2:
                    print('Index out of list. Exiting with acquisition function value {} at {}'.format(
                        val_list[pos], idx_list[pos]))
                break
            if self.verbose == 2:
                print('Acquisition function max value {} at {}'.format(val_list[pos], idx_list[pos]))
        return idx_list[pos], val_list[pos]

    # ------------------------------------------------------------------ loop
    def single_step(self, *args):
        e = args[0]
        if self.verbose:
            print("\nExploration step {} / {}".format(e + 1, self.exploration_steps))
        if e == 0:
            self.surrogate_model.train()
        vals, inds = self.next_point()
        if not self.batch_update:
            inds, vals = self.checkvalues(inds, vals)
        self.evaluate_function(inds)
        self.update_posterior()
        if isinstance(vals, float):
            self.indices_all.append(inds)
            self.vals_all.append(vals)
        else:
            self.indices_all.extend(inds)
            self.vals_all.extend(vals)

    def run(self):
        for i in range(self.exploration_steps):
            self.single_step(i)
            if self.save_checkpoints:
                self.save_results()
        self.save_results()
        if self.verbose:
            print("\nExploration completed")

    def save_results(self, *args):
        """np.save of {'gp_pred','func_val','inds_all','vals_all'} (boptim.py:472-485)."""
        filename = args[0] if args else self.filename
        results = {'gp_pred': list(self.gp_predictions), 'func_val': self.target_func_vals,
                   'inds_all': np.array(self.indices_all), 'vals_all': np.array(self.vals_all)}
        np.save(filename + ".npy", results)
