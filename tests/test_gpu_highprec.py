"""The HIP engine against the independent 50-digit known answers of tests/golden/gp_highprec.npz
(Matern52 / RationalQuadratic / RBF exact GPs and the sparse VFE model): not via the oracle."""
import ctypes

import numpy as np
import pytest
import torch
from numpy.testing import assert_allclose

pytestmark = pytest.mark.gpu

from test_oracle_highprec import load_cases


def test_hip_vs_high_precision(ensure_built, golden_dir):
    from gpim_amd import _lib
    from gpim_amd.kernels import KernelSpec
    H = _lib.Handle()
    for c in load_cases(golden_dir):
        kind, d = str(c["kind"]), c["X"].shape[1]
        spec = KernelSpec(kind, d, [c["ls"][0].tolist(), c["ls"][1].tolist()], jitter=float(c["jitter"]))
        m = spec.struct()
        P = spec.n_params
        Xd = torch.from_numpy(c["X"]).cuda().contiguous()
        yd = torch.from_numpy(c["y"]).cuda().contiguous()
        ud = torch.from_numpy(c["u"]).cuda().contiguous()
        Xs = torch.from_numpy(c["Xs"]).cuda().contiguous()
        N, M = Xd.shape[0], Xs.shape[0]
        out = torch.empty(1 + P, dtype=torch.float64, device="cuda")
        _lib.check(H.lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                          ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
        o = out.cpu().numpy()
        assert_allclose(o[0], float(c["loss"]), rtol=1e-12)
        assert_allclose(o[1:], c["grad"], rtol=1e-9, atol=1e-11)
        mean = torch.empty(M, dtype=torch.float64, device="cuda")
        var = torch.empty_like(mean)
        _lib.check(H.lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                               _lib.ptr(Xs), M, _lib.ptr(mean), _lib.ptr(var)))
        assert_allclose(mean.cpu().numpy(), c["mean"], rtol=0, atol=1e-11)
        assert_allclose(var.cpu().numpy(), c["var"], rtol=0, atol=1e-11)
        # sparse VFE
        Xu = torch.from_numpy(c["Xu"])
        Mu = Xu.shape[0]
        uv = torch.cat([torch.from_numpy(c["u"]), Xu.reshape(-1)]).cuda().contiguous()
        out = torch.empty(1 + P + Mu * d, dtype=torch.float64, device="cuda")
        _lib.check(H.lib.gpimhip_vfe_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(uv),
                                              ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
        o = out.cpu().numpy()
        assert_allclose(o[0], float(c["vfe_loss"]), rtol=1e-11)
        assert_allclose(o[1:], c["vfe_grad"], rtol=1e-7, atol=1e-9)
        _lib.check(H.lib.gpimhip_predict_vfe(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(uv),
                                             _lib.ptr(Xs), M, _lib.ptr(mean), _lib.ptr(var)))
        assert_allclose(mean.cpu().numpy(), c["vfe_mean"], rtol=0, atol=1e-9)
        assert_allclose(var.cpu().numpy(), c["vfe_var"], rtol=0, atol=1e-9)
    H.close()


@pytest.mark.parametrize("general_path", [False, True])
def test_hip_isotropic_and_4d_vs_high_precision(ensure_built, golden_dir, general_path, monkeypatch):
    """d = 4 and isotropic-lengthscale exact GPs against tests/golden/gp_highprec2.npz, through the fused small-N
    trainer and (GPIMHIP_NO_SMALLN=1) through the general blocked path."""
    from test_oracle_highprec import load_cases2
    from gpim_amd import _lib
    from gpim_amd.kernels import KernelSpec
    if general_path:
        monkeypatch.setenv("GPIMHIP_NO_SMALLN", "1")
    H = _lib.Handle()
    for c in load_cases2(golden_dir):
        kind, d, iso = str(c["kind"]), c["X"].shape[1], bool(c["iso"])
        ls = [float(c["ls"][0]), float(c["ls"][1])] if iso else [c["ls"][0].tolist(), c["ls"][1].tolist()]
        spec = KernelSpec(kind, d, ls, jitter=float(c["jitter"]))
        m = spec.struct()
        P = spec.n_params
        assert P == len(c["u"])
        Xd = torch.from_numpy(c["X"]).cuda().contiguous()
        yd = torch.from_numpy(c["y"]).cuda().contiguous()
        ud = torch.from_numpy(c["u"]).cuda().contiguous()
        Xs = torch.from_numpy(c["Xs"]).cuda().contiguous()
        N, M = Xd.shape[0], Xs.shape[0]
        out = torch.empty(1 + P, dtype=torch.float64, device="cuda")
        _lib.check(H.lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                          ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
        o = out.cpu().numpy()
        assert_allclose(o[0], float(c["loss"]), rtol=1e-12)
        assert_allclose(o[1:], c["grad"], rtol=1e-9, atol=1e-11)
        mean = torch.empty(M, dtype=torch.float64, device="cuda")
        var = torch.empty_like(mean)
        _lib.check(H.lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                               _lib.ptr(Xs), M, _lib.ptr(mean), _lib.ptr(var)))
        assert_allclose(mean.cpu().numpy(), c["mean"], rtol=0, atol=1e-11)
        assert_allclose(var.cpu().numpy(), c["var"], rtol=0, atol=1e-11)
    H.close()
