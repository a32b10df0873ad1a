"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/gpimhip.h declares; the Python binding types all of them (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpimhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpimhip_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "gpimhip_fit_exact" in syms and "gpimhip_predict_exact" in syms and len(syms) >= 12


def test_library_exports_header(ensure_built):
    lib = ctypes.CDLL(ensure_built)
    for s in declared_symbols():
        assert hasattr(lib, s), "libgpimhip.so does not export %s" % s


def test_binding_covers_header(ensure_built):
    from gpim_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_symbols()
    lib = _lib.load()
    assert lib.gpimhip_version() >= 100
    assert ctypes.sizeof(_lib.ModelStruct) == 4 * 4 + 2 * 8 + 2 * 4 * 8 + 8


def test_product_fails_loudly_without_gpu(ensure_built):
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gpim_amd
    R = np.random.rand(8, 8)
    R[::2, ::2] = np.nan
    with pytest.raises(RuntimeError):
        gpim_amd.reconstructor(gpim_amd.utils.get_sparse_grid(R), R, gpim_amd.utils.get_full_grid(R))


def test_gpim_alias_package():
    """``import gpim`` resolves to the engine under the reference's own names (gpim/__init__.py:1-5
    and the sub-module paths user code imports from)."""
    import gpim
    import gpim_amd
    assert gpim.reconstructor is gpim_amd.reconstructor
    assert gpim.boptimizer is gpim_amd.boptimizer
    assert gpim.utils.get_sparse_grid is gpim_amd.utils.get_sparse_grid
    from gpim.gpreg.gpr import reconstructor
    from gpim.gpbayes.boptim import boptimizer
    from gpim.gpbayes import acqfunc
    from gpim.kernels.pyro_kernels import get_kernel
    from gpim import gprutils
    assert reconstructor is gpim_amd.reconstructor and boptimizer is gpim_amd.boptimizer
    assert acqfunc.expected_improvement is gpim_amd.acqfunc.expected_improvement
    assert get_kernel("RBF", 2, [[0., 0.], [5., 5.]]).n_params == 4
    assert gprutils.get_full_grid is gpim_amd.utils.get_full_grid
    import pytest
    assert gpim.skreconstructor is gpim_amd.skreconstructor
    with pytest.raises(NotImplementedError):
        gpim.vreconstructor()
