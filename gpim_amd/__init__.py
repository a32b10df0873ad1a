"""
gpim_amd -- MI355X-native engine for GPim's exact-GP reconstruction / Bayesian-optimisation
hot path.  Same public names as the reference package root (gpim/__init__.py:1-5) for the
part that is in scope: ``reconstructor``, ``boptimizer``, ``utils``.
"""
from . import gprutils as utils
from .gpr import reconstructor
from .skgpr import skreconstructor
from .boptim import boptimizer
from . import acqfunc

__all__ = ["reconstructor", "skreconstructor", "boptimizer", "utils", "acqfunc"]
__version__ = "0.1.0"
