"""``gpim.gpbayes.boptim`` -> gpim_amd.boptim (reference: gpim/gpbayes/boptim.py:22-485)."""
from gpim_amd.boptim import boptimizer            # noqa: F401
