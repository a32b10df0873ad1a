"""The chain-stream hop (factor_at_u) against GPIMHIP_NO_CHAIN_STREAM=1: same bits, both precisions' engines untouched otherwise.
    python tools/r3_chain_check.py [N] [T]"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
side = int(np.ceil(np.sqrt(N * 4))); rng = np.random.default_rng(0)
flat = rng.choice(side * side, size=N, replace=False); flat.sort()
X = np.stack([flat // side, flat % side], 1).astype(np.float64)
y = np.sin(X[:, 0] / 7.0) * np.cos(X[:, 1] / 5.0) + 0.05 * rng.standard_normal(N)
Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5)
torch.manual_seed(0); u0 = spec.draw_initial_u().to(dev); m = spec.struct()
for prec in ("double", "single"):
    res = {}
    for mode in ("chain", "caller", "chain"):
        if mode == "caller": os.environ["GPIMHIP_NO_CHAIN_STREAM"] = "1"
        else: os.environ.pop("GPIMHIP_NO_CHAIN_STREAM", None)
        H = _lib.Handle(precision=prec); lib = H.lib
        u = u0.clone(); hist = torch.zeros(T, spec.n_params, dtype=torch.float64, device=dev); loss = torch.zeros(T, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, T, _lib.ptr(hist), _lib.ptr(loss)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.setdefault(mode, []).append((hist.cpu().numpy().copy(), loss.cpu().numpy().copy(), dt / T * 1e3))
        H.close()
    c, k = res["chain"][0], res["caller"][0]
    print("%s N=%d T=%d: history bitwise equal %s, loss bitwise equal %s, finite %s, repeatable %s | ms/iter chain %.2f / %.2f, caller's stream %.2f" % (
        prec, N, T, np.array_equal(c[0], k[0]), np.array_equal(c[1], k[1]), bool(np.isfinite(c[0]).all() and np.isfinite(c[1]).all()),
        np.array_equal(c[0], res["chain"][1][0]), c[2], res["chain"][1][2], k[2]), flush=True)
