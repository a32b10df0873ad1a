"""Per-call cost of the captured large-N iteration: eager vs captured on ONE warm handle, T iterations each.
    GPIMHIP_GRAPH_PROFILE=1 python tools/r3_graph_large_cost.py T N [N ...]"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
T = int(sys.argv[1]); dev = torch.device("cuda:0")
for N in [int(a) for a in sys.argv[2:]]:
    side = int(np.ceil(np.sqrt(N * 4))); rng = np.random.default_rng(0)
    flat = rng.choice(side * side, size=N, replace=False); flat.sort()
    X = np.stack([flat // side, flat % side], 1).astype(np.float64)
    y = np.sin(X[:, 0] / 7.0) * np.cos(X[:, 1] / 5.0) + 0.05 * rng.standard_normal(N)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    spec = KernelSpec("Matern52", 2, [[1., 1.], [20., 20.]], jitter=1e-5)
    torch.manual_seed(0); u0 = spec.draw_initial_u().to(dev); m = spec.struct()
    H = _lib.Handle(); lib = H.lib
    hist = torch.zeros(T, spec.n_params, dtype=torch.float64, device=dev)
    def fit(mode, t):
        if mode == "eager": os.environ.pop("GPIMHIP_GRAPH_LARGE", None)
        else: os.environ["GPIMHIP_GRAPH_LARGE"] = "1"
        u = u0.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, t, _lib.ptr(hist), None))
        torch.cuda.synchronize(); return time.perf_counter() - t0
    fit("eager", 8); fit("graph", 8)                       # workspace, plans, streams, first graph of the process
    r = {k: [] for k in ("eager", "graph")}
    for rep in range(2):
        for mode in ("eager", "graph"):
            r[mode].append(fit(mode, T))
    e, g = min(r["eager"]), min(r["graph"])
    print("N=%5d T=%d: eager %.1f ms (%.3f ms/iter), captured %.1f ms (%.3f ms/iter): per-call overhead %+.1f ms = %+.2f %%   [all: eager %s graph %s]"
          % (N, T, e * 1e3, e / T * 1e3, g * 1e3, g / T * 1e3, (g - e) * 1e3, (g - e) / e * 100,
             ["%.0f" % (x * 1e3) for x in r["eager"]], ["%.0f" % (x * 1e3) for x in r["graph"]]), flush=True)
    H.close()
