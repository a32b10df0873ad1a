"""
``gpim`` -- drop-in alias of the MI355X engine under the reference package's own name.

A script written for ziatdinovmax/GPim (``import gpim; gpim.reconstructor(...).run()``,
``gpim.boptimizer(...).run()``, ``gpim.utils.get_sparse_grid(...)``) runs unchanged with this
directory on ``sys.path``: the names of the reference's package root that are on the hot path
(gpim/__init__.py:1-5: ``utils``, ``reconstructor``, ``boptimizer``) and the sub-module paths that
user code imports from (``gpim.gpreg.gpr``, ``gpim.gpbayes.boptim``, ``gpim.gpbayes.acqfunc``,
``gpim.kernels.pyro_kernels``, ``gpim.gprutils``) resolve to ``gpim_amd``.  ``skreconstructor``
resolves to the engine's exact Kronecker-structured reconstructor (gpim_amd/skgpr.py); the GPyTorch-based
vector-valued ``vreconstructor`` is outside the scope of this engine and raises
NotImplementedError when called.
"""
from gpim_amd import gprutils as utils            # noqa: F401
from gpim_amd import gprutils                     # noqa: F401  (``from gpim import gprutils``)
from gpim_amd.gpr import reconstructor            # noqa: F401
from gpim_amd.boptim import boptimizer            # noqa: F401
from gpim_amd import __version__                  # noqa: F401


def _out_of_scope(name):
    def ctor(*args, **kwargs):
        raise NotImplementedError(
            "gpim.%s (GPyTorch structured-kernel / vector-valued GP) is outside the scope of the MI355X "
            "engine; use gpim.reconstructor (exact or sparse=True)" % name)
    ctor.__name__ = name
    return ctor


from gpim_amd.skgpr import skreconstructor        # noqa: F401,E402  (exact Kronecker solver in the SKI class's role)

vreconstructor = _out_of_scope("vreconstructor")

__all__ = ["utils", "reconstructor", "boptimizer", "skreconstructor", "vreconstructor"]
