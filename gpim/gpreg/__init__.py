from .gpr import reconstructor                    # noqa: F401
