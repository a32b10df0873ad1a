"""Single-precision engine against the double-precision one on the same inputs (loss, gradient, posterior) and
its speed.  usage: fp32_probe.py N [N ...]"""
import ctypes, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
dev = torch.device("cuda:0")
H64, H32 = _lib.Handle(), _lib.Handle(precision="single")
for N in [int(a) for a in sys.argv[1:]]:
    side = int(np.ceil(np.sqrt(N * 4)))
    rng = np.random.default_rng(0)
    flat = rng.choice(side * side, size=N, replace=False); flat.sort()
    X = np.stack([flat // side, flat % side], 1).astype(np.float64)
    y = np.sin(X[:, 0] / 7.0) * np.cos(X[:, 1] / 5.0) + 0.05 * rng.standard_normal(N)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    torch.manual_seed(0)
    spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5)
    u0 = spec.draw_initial_u().to(dev); m = spec.struct()
    M = 4096
    Xs = torch.from_numpy(rng.uniform(0, side, size=(M, 2))).to(dev)
    res = {}
    for name, H in (("f64", H64), ("f32", H32)):
        lib = H.lib
        out = torch.empty(1 + spec.n_params, dtype=torch.float64, device=dev)
        _lib.check(lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u0),
                                        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
        mean = torch.empty(M, dtype=torch.float64, device=dev); var = torch.empty_like(mean)
        _lib.check(lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u0), _lib.ptr(Xs), M,
                                             _lib.ptr(mean), _lib.ptr(var)))
        T = 6
        u = u0.clone(); hist = torch.empty(T, spec.n_params, dtype=torch.float64, device=dev)
        for rep in range(2):
            u.copy_(u0); torch.cuda.synchronize(); t = time.time()
            _lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, T, _lib.ptr(hist), None))
            torch.cuda.synchronize(); dt = (time.time() - t) / T
        res[name] = (out.cpu().numpy(), mean.cpu().numpy(), var.cpu().numpy(), u.cpu().numpy(), dt)
        print("N=%d %s: loss %.10g  grad %s  %.3f ms/iter (%.1f TFLOP/s)  ws %.2f GiB" % (N, name, res[name][0][0], np.array2string(res[name][0][1:], precision=5), dt * 1e3, N ** 3 / dt / 1e12, lib.gpimhip_workspace_bytes(H.h) / 2 ** 30), flush=True)
    a, b = res["f64"], res["f32"]
    print("   f32 vs f64: loss rel %.2e  grad rel %.2e  mean max abs %.2e  var max rel %.2e  u(T) max abs %.2e  speedup %.2fx" % (
        abs(b[0][0] - a[0][0]) / abs(a[0][0]), np.abs(b[0][1:] - a[0][1:]).max() / np.abs(a[0][1:]).max(),
        np.abs(b[1] - a[1]).max(), (np.abs(b[2] - a[2]) / a[2]).max(), np.abs(b[3] - a[3]).max(), a[4] / b[4]), flush=True)
