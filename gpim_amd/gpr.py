"""
gpr.py -- ``reconstructor``: exact Gaussian-process regression on image / hyperspectral grids,
computed by the MI355X engine (libgpimhip.so).

Host-side mirror of the reference's gpim/gpreg/gpr.py:22-283:
same constructor signature, same ``train`` / ``predict`` / ``run`` methods and return values,
same ``hyperparams`` dictionary.  What the reference delegates to pyro.contrib.gp / torch
(kernel matrix, Cholesky, marginal log-likelihood and its gradient, Adam, posterior) runs as
hand-written HIP kernels behind the C ABI of include/gpimhip.h.

Deliberate differences from the reference, all on the outside of the numerics:
  * the engine is GPU-only.  ``use_gpu`` is accepted for signature compatibility and ignored;
    without a HIP device or without libgpimhip.so every call raises RuntimeError.
  * the initial hyper-parameters are always drawn with torch's CPU generator (the reference does
    that on its CPU path; its CUDA path uses the CUDA generator and is not reproducible).
  * no process-global side effects: torch's default tensor type is left alone.
  * a non-positive-definite covariance during training freezes the hyper-parameters, the optimiser
    and the history on the device at the failing iteration (what the reference's state is when
    torch.linalg.cholesky raises, gpr.py:192); the host stops enqueueing a bounded number of
    iterations later and ``train`` raises the same exception class with the history up to that
    iteration appended, so the reconstructor stays usable.
  * ``precision='single'``: inputs are taken, the initial hyper-parameters drawn and the results
    returned in float32 like the reference does, and the exact-GP engine runs in single precision
    (gpimhip_set_precision(h, 32): float N x N matrices, every O(N^3) product on the fp32 matrix
    cores -- 1.7x the double-precision throughput at N = 16384 and half the memory); diagonal blocks,
    the O(N) vectors, loss, gradient and Adam stay double and K^-1 y is refined once against the
    covariance in double, so posterior means are at least as accurate as a float32 run of the reference
    (variances and the loss: float32 round-off times the conditioning), not bit-comparable to one.  Sparse and structured models, and the
    fused trainer for N <= 128, compute in double on either setting.
"""
import ctypes
import random
import time
import warnings

import numpy as np
import torch

from . import _lib, gprutils
from .kernels import get_kernel

_F64 = torch.float64


class _KernelView:
    """``model.kernel`` facade: constrained values as tensors (boptim.py:319 reads
    ``model.kernel.lengthscale.mean().item()``)."""

    def __init__(self, owner):
        self._o = owner

    @property
    def variance(self):
        return self._o._spec.constrained(self._o._u)[0]

    @property
    def lengthscale(self):
        ls = self._o._spec.constrained(self._o._u)[1]
        return ls.reshape(()) if self._o._spec.isotropic else ls

    variance_map = variance
    lengthscale_map = lengthscale


class _ModelView:
    """``reconstructor.model`` facade with settable training data (boptim.py:248-249 swaps
    ``model.X`` / ``model.y`` in place between trainings)."""

    def __init__(self, owner):
        self._o = owner
        self.kernel = _KernelView(owner)

    @property
    def X(self):
        return self._o._Xd

    @X.setter
    def X(self, value):
        self._o._no_swap("X")
        self._o._Xd = self._o._to_device(value)

    @property
    def y(self):
        return self._o._yd

    @y.setter
    def y(self, value):
        self._o._no_swap("y")
        self._o._yd = self._o._to_device(value)

    @property
    def noise(self):
        return self._o._spec.constrained(self._o._u)[2]

    @property
    def jitter(self):
        return self._o._spec.jitter

    @property
    def Xu(self):
        o = self._o
        P = o._spec.n_params
        return o._u[P:].reshape(o._n_ind, o._spec.dim) if o._n_ind else None

    def parameters(self):
        yield self._o._u


class reconstructor:
    """
    Gaussian-process reconstruction of sparse 2D images and 3D/4D hyperspectral data.

    Args:
        X (ndarray): grid indices, :math:`c \\times N \\times M (\\times L ...)`; NaN = missing
        y (ndarray): observations, :math:`N \\times M (\\times L ...)`; NaN = missing
        Xtest (ndarray): grid on which to predict (same layout as X)
        kernel (str): 'RBF', 'Matern52' or 'RationalQuadratic'
        lengthscale: ``[lo, hi]`` (one shared lengthscale) or ``[[lo...], [hi...]]`` bounds;
            default ``[[0]*d, [mean(y.shape)/2]*d]``
        sparse (bool), indpoints (int): sparse variational GP (VFE) with trainable inducing inputs, initialised as
            X[::len(X) // indpoints] like the reference (csrc/vfe.hip)
        learning_rate (float), iterations (int): Adam settings
        use_gpu: ignored (always on the GPU)
        verbose (int): 0, 1 or 2
        seed (int): seeds the CPU generator used for the initial hyper-parameter draw
        **amplitude, **precision, **jitter, **isotropic: as in the reference
        **structured (bool): gpim_amd extension -- exact GP through the Kronecker structure of the
            covariance (csrc/kron.hip).  Requires fully observed data on a product grid (X as returned
            by ``utils.get_full_grid``) and the RBF kernel; same model, parameterisation and results as
            the dense path, O(sum n_i^3 + N sum n_i) instead of O(N^3) work.  Takes the role of the
            reference's structured-kernel ``skreconstructor`` (gpim/gpreg/skgpr.py), exact rather than
            interpolated.  With a kernel that does not factorise over the axes ('Matern52', 'RationalQuadratic') the
            reflection symmetry of the complete grid is used instead: in the basis adapted to the reflections of every
            axis the covariance is block diagonal -- 2^r dense blocks of N / 2^r points, one lock-step
            batch with shared hyper-parameters (csrc/engine.hip: kmat_refl_kernel) -- 4^-r of the dense model's O(N^3)
            work (1/16 for a 2-D image), the same model and results to rounding; predictions at arbitrary points.  Axes of
            odd length are reflected too (their mirror plane belongs to the fundamental domain, with weights).
    """

    def __init__(self, X, y, Xtest=None, kernel='RBF', lengthscale=None, sparse=False,
                 indpoints=None, learning_rate=5e-2, iterations=1000, use_gpu=False,
                 verbose=1, seed=0, **kwargs):
        self.precision = kwargs.get("precision", "double")
        self._np_out = np.float32 if self.precision == "single" else np.float64
        # raises if there is no GPU / no library; single precision applies to the exact dense engine only
        single_engine = self.precision == "single" and not sparse and not kwargs.get("structured", False)
        self._handle = _lib.Handle(precision="single" if single_engine else "double")
        self._dev = self._handle.device
        self.verbose = verbose
        # pyro.set_rng_seed(seed) at gpr.py:101 seeds torch, numpy and Python's random: the boptimizer
        # paths that draw from np.random (checkvalues' exit strategy, update_points' padding) depend on it
        torch.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)
        input_dim = np.ndim(y)
        self.X, self.y = gprutils.prepare_training_data(X, y, precision=self.precision)
        self.do_sparse = bool(sparse)
        self.do_structured = bool(kwargs.get("structured", False))
        self.do_symm = False
        self._kernel_name = kernel
        if self.do_structured:
            if self.do_sparse:
                raise NotImplementedError("structured=True and sparse=True are mutually exclusive")
            if np.isnan(np.asarray(y)).any():
                raise NotImplementedError("structured=True needs a fully observed grid (no NaN in y)")
            self._axes, self._axes_n = self._grid_axes(X)
            if kernel != "RBF":
                # kernels that do not factorise over the axes (Matern52, RationalQuadratic): the reflection symmetry of the
                # complete grid instead -- 2^r diagonal blocks of N / 2^r points each, r = the axes of even length
                self.do_structured = False
                self.do_symm = True
                self._symm_setup(np.asarray(X, dtype=np.float64), np.asarray(y, dtype=np.float64))
        if lengthscale is None and not kwargs.get("isotropic"):
            lmean = float(np.mean(y.shape) / 2)
            lengthscale = [[0. for _ in range(input_dim)], [lmean for _ in range(input_dim)]]
        elif lengthscale is None and kwargs.get("isotropic"):
            lengthscale = [0., float(np.mean(y.shape) / 2)]
        self._spec = get_kernel(kernel, input_dim, lengthscale, use_gpu,
                                amplitude=kwargs.get('amplitude'), precision=self.precision,
                                jitter=kwargs.get("jitter", 1.0e-5))
        # drawn from a private generator seeded like the global one: identical numbers, no race when
        # several reconstructors are built from different threads
        self._u = self._spec.draw_initial_u(
            torch.Generator().manual_seed(seed),
            torch.float32 if self.precision == "single" else _F64).to(self._dev)
        self._mstruct = self._spec.struct()
        self._n_ind = 0
        if self.do_sparse:
            # inducing inputs: every (N // indpoints)-th observation, trainable (gpr.py:145-153)
            n = len(self.X)
            if indpoints is None:
                indpoints = n // 10
                indpoints = indpoints + 1 if indpoints == 0 else indpoints
            else:
                indpoints = n if indpoints > n else indpoints
            Xu = self.X[::n // indpoints]
            if self.verbose == 2:
                print("# of inducing points for sparse GP regression: {}".format(len(Xu)))
            self._n_ind = len(Xu)
            self._u = torch.cat([self._u, Xu.reshape(-1).to(self._dev, _F64)]).contiguous()
        self.fulldims = Xtest.shape[1:] if Xtest is not None else X.shape[1:]
        self.Xtest = gprutils.prepare_test_data(Xtest, precision=self.precision) if Xtest is not None else None
        self._Xd = self._to_device(self.X)
        self._yd = self._to_device(self.y)
        self._Xtest_d = self._to_device(self.Xtest) if self.Xtest is not None else None
        if self.do_structured:
            self._axes_d = self._to_device(np.concatenate(self._axes))
            self._taxes = self._grid_axes(Xtest) if Xtest is not None else (self._axes, self._axes_n)
        self.learning_rate = learning_rate
        self.iterations = iterations
        self.indpoints_all = []
        self.lscales, self.noise_all, self.amp_all = [], [], []
        self.loss_all = []
        self.hyperparams = {
            "lengthscale": self.lscales,
            "noise": self.noise_all,
            "variance": self.amp_all,
            "inducing_points": self.indpoints_all
        }
        self._last_pred = None       # (mean, sd) device tensors of the latest predict()

    @property
    def model(self):
        """The ``model`` facade of the reference object (``model.X`` / ``model.y`` settable, ``model.kernel``).  Built per
        access: a stored view would close a reference cycle with this object, and the library handle -- with its N x N
        workspaces -- would then live until the cyclic collector happens to run (seen as a 50-65 ms hipFree inside the NEXT
        Bayesian-optimisation run's first training, tools/r5_c4_steps.py)."""
        return _ModelView(self)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _grid_axes(X):
        """Coordinate vectors of a product grid X (d, n_1, ..., n_d): X[i] must vary along axis i only
        (what utils.get_full_grid returns, with or without ``extent`` / ``dense_x``)."""
        X = np.asarray(X, dtype=np.float64)
        d = X.shape[0]
        if X.ndim != d + 1:
            raise NotImplementedError("structured=True needs grid coordinates of shape (d, n_1, ..., n_d)")
        axes = []
        for i in range(d):
            c = np.moveaxis(X[i], i, 0).reshape(X.shape[1 + i], -1)[:, 0].copy()
            shape = [1] * d
            shape[i] = -1
            if not np.array_equal(X[i], np.broadcast_to(c.reshape(shape), X.shape[1:])):
                raise NotImplementedError("structured=True needs a product grid (coordinate i varying along axis i only)")
            axes.append(c)
        return axes, (ctypes.c_int32 * d)(*[len(c) for c in axes])

    def _symm_setup(self, X, y):
        """The symmetry-reduced form of the model (gprutils.reflection_blocks; csrc/engine.hip: kmat_refl_kernel)."""
        try:
            S = gprutils.reflection_blocks(X, y, self._axes)
        except ValueError as e:
            raise NotImplementedError("structured=True with kernel %r: %s" % (self._kernel_name, e))
        S["twoc"] = (ctypes.c_double * 4)(*S["twoc"])
        self._symm = S

    def _symm_call(self, fn, var_count=0):
        """fn(Xq, ys, Nq, B, u_b) with the handle in reflection mode; the B parameter slots hold one vector."""
        S, lib, h = self._symm, self._handle.lib, self._handle.h
        if "Xq_d" not in S:
            S["Xq_d"], S["ys_d"] = self._to_device(S["Xq"]), self._to_device(S["ys"])
            S["wts_d"] = self._to_device(S["wts"]) if S["wts"] is not None else None
        u_b = self._u.repeat(S["B"]).contiguous()
        _lib.check(lib.gpimhip_set_reflection(h, S["mask"], S["twoc"], _lib.ptr(S["wts_d"]), S["n_total"], int(var_count)))
        try:
            rc = fn(S["Xq_d"], S["ys_d"], S["Xq_d"].shape[0], S["B"], u_b)
        finally:
            _lib.check(lib.gpimhip_set_reflection(h, 0, None, None, 0, 0))
        self._u.copy_(u_b[:self._u.numel()])
        return rc

    def _no_swap(self, what):
        """structured=True models are built from the complete grid once (coordinate vectors of the Kronecker solver, the
        reflection blocks and projected observations of the symmetry-reduced one): swapping the training data through
        ``model.X`` / ``model.y`` would leave them stale and train / predict on the old observations silently."""
        if self.do_symm or self.do_structured:
            raise NotImplementedError("structured=True: the training data cannot be replaced through model.%s "
                                      "(build a new reconstructor for new observations)" % what)

    def _to_device(self, t):
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        return t.detach().to(self._dev, _F64).contiguous()

    def _check_data(self):
        X, y = self._Xd, self._yd
        if X.dim() != 2 or X.shape[1] != self._spec.dim or y.dim() != 1 or y.shape[0] != X.shape[0]:
            raise ValueError("training data must be X:(N,%d), y:(N,); got %s and %s"
                             % (self._spec.dim, tuple(X.shape), tuple(y.shape)))
        if X.shape[0] < 1:
            raise ValueError("no observations (all NaN)")

    # ------------------------------------------------------------------ training
    def train(self, **kwargs):
        """Adam on the negative log marginal likelihood; every call starts a fresh optimiser
        while the hyper-parameters persist (warm start), like gpr.py:170-217."""
        if kwargs.get("learning_rate") is not None:
            self.learning_rate = kwargs.get("learning_rate")
        if kwargs.get("iterations") is not None:
            self.iterations = kwargs.get("iterations")
        if kwargs.get("verbose") is not None:
            self.verbose = kwargs.get("verbose")
        self._check_data()
        T, P = int(self.iterations), self._spec.n_params
        start_time = time.time()
        if self.verbose:
            print('Model training...')
        hist = torch.empty((max(T, 1), P), dtype=_F64, device=self._dev)
        loss = torch.empty((max(T, 1),), dtype=_F64, device=self._dev)
        if self.do_structured:
            rc = self._handle.lib.gpimhip_fit_kron(
                self._handle.h, ctypes.byref(self._mstruct), self._spec.dim, self._axes_n, _lib.ptr(self._axes_d),
                _lib.ptr(self._yd), _lib.ptr(self._u), float(self.learning_rate), T, _lib.ptr(hist), _lib.ptr(loss))
        elif self.do_symm:
            Bs = self._symm["B"]
            hist_b = torch.empty((Bs, max(T, 1), P), dtype=_F64, device=self._dev)
            loss_b = torch.empty((Bs, max(T, 1)), dtype=_F64, device=self._dev)
            rc = self._symm_call(lambda Xq, ys, Nq, B, u_b: self._handle.lib.gpimhip_fit_exact_batched(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(Xq), 0, _lib.ptr(ys), Nq, B, _lib.ptr(u_b),
                float(self.learning_rate), T, _lib.ptr(hist_b), _lib.ptr(loss_b)))
            hist, loss = hist_b[0], loss_b[0]
        elif not self.do_sparse:
            rc = self._handle.lib.gpimhip_fit_exact(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(self._Xd), _lib.ptr(self._yd),
                self._Xd.shape[0], _lib.ptr(self._u), float(self.learning_rate), T,
                _lib.ptr(hist), _lib.ptr(loss))
        else:
            d = self._spec.dim
            hist_xu = torch.empty((max(T, 1), self._n_ind, d), dtype=_F64, device=self._dev)
            rc = self._handle.lib.gpimhip_fit_vfe(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(self._Xd), _lib.ptr(self._yd),
                self._Xd.shape[0], self._n_ind, _lib.ptr(self._u), float(self.learning_rate), T,
                _lib.ptr(hist), _lib.ptr(hist_xu), _lib.ptr(loss))
        failed = rc == _lib.E_NOT_PD
        if failed:
            # the device loop froze the parameters at the failing iteration: keep the history up to it
            # and raise what torch.linalg.cholesky raises there in the reference (gpr.py:192)
            T = int(self._handle.lib.gpimhip_fit_completed(self._handle.h))
        else:
            _lib.check(rc)
        if self.do_sparse and T > 0:
            self.indpoints_all.extend(list(hist_xu[:T].cpu().numpy()))
        hist_h = hist[:T].cpu().numpy()
        loss_h = loss[:T].cpu().numpy()
        n_ls = self._spec.n_ls
        for i in range(T):
            row = hist_h[i]
            self.lscales.append(float(row[1]) if self._spec.isotropic else row[1:1 + n_ls].tolist())
            self.amp_all.append(float(row[0]))
            self.noise_all.append(float(row[1 + n_ls]))
            self.loss_all.append(float(loss_h[i]))
            if self.verbose == 2 and (i % 100 == 0 or i == T - 1):
                print('iter: {} ...'.format(i),
                      'loss: {} ...'.format(np.around(loss_h[i], 4)),
                      'amp: {} ...'.format(np.around(self.amp_all[-1], 4)),
                      'length: {} ...'.format(np.around(self.lscales[-1], 4)),
                      'noise: {} ...'.format(np.around(self.noise_all[-1], 7)))
        if failed:
            _lib.check(rc)
        if self.verbose:
            dt = time.time() - start_time
            if T > 0:
                print('average time per iteration: {} s'.format(np.round(dt / T, 6)))
            print('training completed in {} s'.format(np.round(dt, 2)))
            var, ls, noise = self._spec.constrained(self._u)
            print('Final parameter values:\n',
                  'amp: {}, lengthscale: {}, noise: {}'.format(
                      np.around(var.item(), 4), np.around(ls.tolist(), 4), np.around(noise.item(), 7)))
        return

    # ------------------------------------------------------------------ prediction
    def predict(self, Xtest=None, **kwargs):
        """Posterior mean and standard deviation (noise included) on the test grid;
        returns numpy arrays shaped like the grid (gpr.py:219-255)."""
        if Xtest is None and self.Xtest is None:
            warnings.warn("No test data provided. Using training data for prediction", UserWarning)
            self.Xtest = self.X
            self._Xtest_d = self._Xd
        elif Xtest is not None:
            self.Xtest = gprutils.prepare_test_data(Xtest, precision=self.precision)
            self._Xtest_d = self._to_device(self.Xtest)
            self.fulldims = Xtest.shape[1:]
            if self.do_structured:
                self._taxes = self._grid_axes(Xtest)
        if kwargs.get("verbose") is not None:
            self.verbose = kwargs.get("verbose")
        if self.verbose:
            print("Calculating predictive mean and variance...", end=" ")
        self._check_data()
        M = self._Xtest_d.shape[0]
        mean = torch.empty((M,), dtype=_F64, device=self._dev)
        var = torch.empty((M,), dtype=_F64, device=self._dev)
        if self.do_structured:
            taxes, tn = self._taxes
            if int(np.prod([len(c) for c in taxes])) != M:
                raise NotImplementedError("structured=True predicts on product grids only")
            rc = self._handle.lib.gpimhip_predict_kron(
                self._handle.h, ctypes.byref(self._mstruct), self._spec.dim, self._axes_n, _lib.ptr(self._axes_d),
                _lib.ptr(self._yd), _lib.ptr(self._u), tn, _lib.ptr(self._to_device(np.concatenate(taxes))),
                _lib.ptr(mean), _lib.ptr(var))
        elif self.do_symm:
            S = self._symm
            on_grid = self._Xtest_d.shape == self._Xd.shape and bool(torch.equal(self._Xtest_d, self._Xd))
            if on_grid:
                # the training grid itself: the variance is invariant under the reflections -- computed on the fundamental
                # domain (ordered first) and mirrored; the mean everywhere
                if "perm_d" not in S:
                    rest = np.setdiff1d(np.arange(M), S["fund_flat"], assume_unique=True)
                    S["perm_d"] = torch.from_numpy(np.concatenate([S["fund_flat"], rest])).to(self._dev)
                    S["rep_d"] = torch.from_numpy(S["rep"]).to(self._dev)
                Xt = self._Xtest_d[S["perm_d"]].contiguous()
                mean_p = torch.empty((M,), dtype=_F64, device=self._dev)
                var_p = torch.empty((M,), dtype=_F64, device=self._dev)
                nq = len(S["fund_flat"])
                rc = self._symm_call(lambda Xq, ys, Nq, B, u_b: self._handle.lib.gpimhip_predict_exact_batched(
                    self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(Xq), 0, _lib.ptr(ys), Nq, B, _lib.ptr(u_b),
                    _lib.ptr(Xt), M, _lib.ptr(mean_p), _lib.ptr(var_p)), var_count=nq)
                mean[S["perm_d"]] = mean_p
                var = var_p[:nq][S["rep_d"]]
            else:
                rc = self._symm_call(lambda Xq, ys, Nq, B, u_b: self._handle.lib.gpimhip_predict_exact_batched(
                    self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(Xq), 0, _lib.ptr(ys), Nq, B, _lib.ptr(u_b),
                    _lib.ptr(self._Xtest_d), M, _lib.ptr(mean), _lib.ptr(var)))
        elif not self.do_sparse:
            rc = self._handle.lib.gpimhip_predict_exact(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(self._Xd), _lib.ptr(self._yd),
                self._Xd.shape[0], _lib.ptr(self._u), _lib.ptr(self._Xtest_d), M, _lib.ptr(mean), _lib.ptr(var))
        else:
            rc = self._handle.lib.gpimhip_predict_vfe(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(self._Xd), _lib.ptr(self._yd),
                self._Xd.shape[0], self._n_ind, _lib.ptr(self._u), _lib.ptr(self._Xtest_d), M, _lib.ptr(mean),
                _lib.ptr(var))
        _lib.check(rc)
        sd = var.sqrt()
        self._last_pred = (mean, sd)
        mean_h = mean.cpu().numpy().reshape(self.fulldims).astype(self._np_out, copy=False)
        sd_h = sd.cpu().numpy().reshape(self.fulldims).astype(self._np_out, copy=False)
        if self.verbose:
            print("Done")
        return mean_h, sd_h

    def _predict_device(self, Xrows_d):
        """Posterior (mean, sd) device tensors at the (M, d) device rows `Xrows_d`; no host copies and no
        change of the stored test grid.  Internal: the device-resident acquisition path of boptimizer."""
        self._check_data()
        M = Xrows_d.shape[0]
        mean = torch.empty((M,), dtype=_F64, device=self._dev)
        var = torch.empty((M,), dtype=_F64, device=self._dev)
        if self.do_structured:
            raise NotImplementedError("structured models predict on product grids (use predict())")
        if self.do_symm:
            # the reflection blocks, not the dense O(N^3) model of the same data
            rc = self._symm_call(lambda Xq, ys, Nq, B, u_b: self._handle.lib.gpimhip_predict_exact_batched(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(Xq), 0, _lib.ptr(ys), Nq, B, _lib.ptr(u_b),
                _lib.ptr(Xrows_d), M, _lib.ptr(mean), _lib.ptr(var)))
        elif not self.do_sparse:
            rc = self._handle.lib.gpimhip_predict_exact(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(self._Xd), _lib.ptr(self._yd),
                self._Xd.shape[0], _lib.ptr(self._u), _lib.ptr(Xrows_d), M, _lib.ptr(mean), _lib.ptr(var))
        else:
            rc = self._handle.lib.gpimhip_predict_vfe(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(self._Xd), _lib.ptr(self._yd),
                self._Xd.shape[0], self._n_ind, _lib.ptr(self._u), _lib.ptr(Xrows_d), M, _lib.ptr(mean),
                _lib.ptr(var))
        _lib.check(rc)
        return mean, var.sqrt()

    def run(self, **kwargs):
        """train + predict; returns (mean, sd, hyperparams) (gpr.py:257-283)."""
        if kwargs.get("learning_rate") is not None:
            self.learning_rate = kwargs.get("learning_rate")
        if kwargs.get("iterations") is not None:
            self.iterations = kwargs.get("iterations")
        self.train(learning_rate=self.learning_rate, iterations=self.iterations)
        mean, sd = self.predict()
        return mean, sd, self.hyperparams

    def step(self, *args, **kwargs):
        """Dead code in the reference (gpr.py:285-329 calls ``gprutils.acquisition``, which does not
        exist, and fails with AttributeError); use ``boptimizer`` for exploration steps."""
        raise AttributeError("module 'gpim.gprutils' has no attribute 'acquisition' "
                             "(reconstructor.step is dead code in the reference; use boptimizer)")

    # ------------------------------------------------------------------ operator-level hooks
    def loss_and_grad(self):
        """(loss, d loss/du) at the current hyper-parameters; used by the parity tests."""
        self._check_data()
        P = self._u.numel()
        out = torch.empty((1 + P,), dtype=_F64, device=self._dev)
        if self.do_structured:
            rc = self._handle.lib.gpimhip_kron_nll_grad(
                self._handle.h, ctypes.byref(self._mstruct), self._spec.dim, self._axes_n, _lib.ptr(self._axes_d),
                _lib.ptr(self._yd), _lib.ptr(self._u), ctypes.c_void_p(out.data_ptr()),
                ctypes.c_void_p(out.data_ptr() + 8))
        elif self.do_sparse:
            rc = self._handle.lib.gpimhip_vfe_nll_grad(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(self._Xd), _lib.ptr(self._yd),
                self._Xd.shape[0], self._n_ind, _lib.ptr(self._u), ctypes.c_void_p(out.data_ptr()),
                ctypes.c_void_p(out.data_ptr() + 8))
        else:
            rc = self._handle.lib.gpimhip_nll_grad(
                self._handle.h, ctypes.byref(self._mstruct), _lib.ptr(self._Xd), _lib.ptr(self._yd),
                self._Xd.shape[0], _lib.ptr(self._u), ctypes.c_void_p(out.data_ptr()),
                ctypes.c_void_p(out.data_ptr() + 8))
        _lib.check(rc)
        o = out.cpu()
        return o[0].item(), o[1:].clone()
