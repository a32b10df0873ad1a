"""``gpim.gpreg.gpr`` -> gpim_amd.gpr (reference: gpim/gpreg/gpr.py:22-283)."""
from gpim_amd.gpr import reconstructor            # noqa: F401
