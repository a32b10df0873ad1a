"""Captured large-N iteration (fit_impl, step schedule) against the eager one: same bits, and the time per iteration.
    python tools/r3_graph_large_check.py [N] [T]"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
side = int(np.ceil(np.sqrt(N * 4))); rng = np.random.default_rng(0)
flat = rng.choice(side * side, size=N, replace=False); flat.sort()
X = np.stack([flat // side, flat % side], 1).astype(np.float64)
y = np.sin(X[:, 0] / 7.0) * np.cos(X[:, 1] / 5.0) + 0.05 * rng.standard_normal(N)
Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5)
torch.manual_seed(0); u0 = spec.draw_initial_u().to(dev); m = spec.struct()
res = {}
for mode in ("eager", "graph", "eager", "graph"):
    if mode == "eager": os.environ.pop("GPIMHIP_GRAPH_LARGE", None)
    else: os.environ["GPIMHIP_GRAPH_LARGE"] = "1"
    H = _lib.Handle(); lib = H.lib
    u = u0.clone(); hist = torch.zeros(T, spec.n_params, dtype=torch.float64, device=dev); loss = torch.zeros(T, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, T, _lib.ptr(hist), _lib.ptr(loss)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.2f ms/iter (incl. capture / first-use costs)" % (mode, dt / T * 1e3), flush=True)
    res.setdefault(mode, []).append((hist.cpu().numpy().copy(), loss.cpu().numpy().copy(), u.cpu().numpy().copy()))
    H.close()
e, g = res["eager"][0], res["graph"][0]
print("N=%d T=%d  history bitwise equal: %s, loss bitwise equal: %s, final u bitwise equal: %s; finite: %s" % (
    N, T, np.array_equal(e[0], g[0]), np.array_equal(e[1], g[1]), np.array_equal(e[2], g[2]), bool(np.isfinite(g[0]).all() and np.isfinite(g[1]).all())))
print("repeatable: eager %s graph %s" % (np.array_equal(res["eager"][0][0], res["eager"][1][0]), np.array_equal(res["graph"][0][0], res["graph"][1][0])))
