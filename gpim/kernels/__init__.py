from . import pyro_kernels                        # noqa: F401
