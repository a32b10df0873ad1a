"""``gpim.kernels.pyro_kernels`` -> gpim_amd.kernels (reference: gpim/kernels/pyro_kernels.py:14-96)."""
from gpim_amd.kernels import get_kernel           # noqa: F401
