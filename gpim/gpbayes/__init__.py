from . import acqfunc                             # noqa: F401
from .boptim import boptimizer                    # noqa: F401
