"""``gpim.gpreg.skgpr`` -> gpim_amd.skgpr (role of the reference's gpim/gpreg/skgpr.py)."""
from gpim_amd.skgpr import skreconstructor        # noqa: F401
