import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
dev = torch.device("cuda:0")
def run(H, N, T=6, M=900):
    lib = H.lib
    side = int(np.ceil(np.sqrt(N * 3)))
    rng = np.random.default_rng(N)
    flat = rng.choice(side * side, size=N, replace=False); flat.sort()
    X = np.stack([flat // side, flat % side], 1).astype(np.float64)
    y = np.sin(X[:, 0] / 7.0) * np.cos(X[:, 1] / 5.0) + 0.05 * rng.standard_normal(N)
    Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
    spec = KernelSpec("RBF", 2, [[1., 1.], [20., 20.]], jitter=1e-5)
    u = spec.draw_initial_u(torch.Generator().manual_seed(0)).to(dev); m = spec.struct()
    hist = torch.empty(T, spec.n_params, dtype=torch.float64, device=dev)
    _lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, T, _lib.ptr(hist), None))
    g = np.stack(np.meshgrid(np.arange(30.), np.arange(30.), indexing="ij"), -1).reshape(-1, 2)
    Xs = torch.from_numpy(g).to(dev)
    mean = torch.empty(len(g), dtype=torch.float64, device=dev); var = torch.empty_like(mean)
    _lib.check(lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), _lib.ptr(Xs), len(g), _lib.ptr(mean), _lib.ptr(var)))
    torch.cuda.synchronize()
    return hist.cpu().numpy(), mean.cpu().numpy(), var.cpu().numpy()
H = _lib.Handle()
ok = True
for N in [1207, 1250, 1207, 1300, 1216, 1217, 1207, 449, 448, 513, 1207]:
    a = run(H, N)
    Hf = _lib.Handle(); b = run(Hf, N); Hf.close()
    same = all(np.array_equal(x, y) for x, y in zip(a, b))
    ok &= same
    print(N, "reused handle == fresh handle:", same, "finite:", np.isfinite(a[1]).all() and np.isfinite(a[2]).all())
print("ALL OK" if ok else "MISMATCH")
