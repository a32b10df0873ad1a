"""
skgpr.py -- ``skreconstructor``: GP reconstruction of fully observed 2D / 3D / 4D grids with a
structured covariance.

Takes the ROLE of the reference's gpim/gpreg/skgpr.py:21-448 (``gpim.skreconstructor``): a reconstructor
for complete images / cubes that exploits the lattice structure of the inputs instead of paying the dense
O(N^3).  The reference does it with GPyTorch's structured kernel interpolation (an approximation, with
GPyTorch's own hyper-parameter parameterisation: constant mean, softplus-constrained scale and noise);
this engine does it EXACTLY -- for the RBF kernel through the Kronecker factorisation of the covariance
(csrc/kron.hip), for 'Matern52' (the reference's other structured kernel, gpim/kernels/gpytorch_kernels.py:65) and
'RationalQuadratic' through the reflection symmetry of the complete grid (2^r dense blocks of N / 2^r points,
gprutils.reflection_blocks + csrc/engine.hip: kmat_refl_kernel) -- with the model and parameterisation of
``gpim_amd.reconstructor`` (zero mean, Uniform priors on variance and lengthscales).  Same constructor shape and return values as the reference class;
numbers are those of ``reconstructor(..., structured=True)``, i.e. of the exact GP -- not bit-comparable
with an SKI run.
"""
from .gpr import reconstructor


class skreconstructor(reconstructor):
    """``skreconstructor(X, y, Xtest=None, kernel='RBF', lengthscale=None, ski=True, learning_rate=.1,
    iterations=50, use_gpu=1, verbose=1, seed=0, **kwargs)`` -- argument order and defaults of
    gpim/gpreg/skgpr.py:79-91.  ``ski``, ``grid_points_ratio``, ``max_root``, ``num_batches`` are accepted
    and ignored (nothing is interpolated or batched); ``kernel``: 'RBF', 'Matern52', 'RationalQuadratic' ('Spectral' is out of
    scope)."""

    def __init__(self, X, y, Xtest=None, kernel='RBF', lengthscale=None, ski=True, learning_rate=.1,
                 iterations=50, use_gpu=1, verbose=1, seed=0, **kwargs):
        for k in ("grid_points_ratio", "max_root", "maxroot", "num_batches", "n_mixtures"):
            kwargs.pop(k, None)
        super().__init__(X, y, Xtest, kernel=kernel, lengthscale=lengthscale, sparse=False, indpoints=None,
                         learning_rate=learning_rate, iterations=iterations, use_gpu=use_gpu, verbose=verbose,
                         seed=seed, structured=True, **kwargs)

    def predict(self, Xtest=None, **kwargs):
        kwargs.pop("num_batches", None)
        kwargs.pop("max_root", None)
        return super().predict(Xtest, **kwargs)
