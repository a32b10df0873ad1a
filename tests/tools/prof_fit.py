"""Times T fit iterations (+1 predict) at a given N; used under rocprofv3 for per-kernel stats.
usage: prof_fit.py N T [M] [kernel]"""
import ctypes, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec

N, T = int(sys.argv[1]), int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else 0
kind = sys.argv[4] if len(sys.argv) > 4 else "RBF"
dev = torch.device("cuda:0")
H = _lib.Handle(precision=os.environ.get("PROF_PRECISION", "double")); lib = H.lib
side = int(np.ceil(np.sqrt(N * 4)))
rng = np.random.default_rng(0)
flat = rng.choice(side * side, size=N, replace=False); flat.sort()
X = np.stack([flat // side, flat % side], 1).astype(np.float64)
y = np.sin(X[:, 0] / 7.0) * np.cos(X[:, 1] / 5.0) + 0.05 * rng.standard_normal(N)
Xd, yd = torch.from_numpy(X).to(dev), torch.from_numpy(y).to(dev)
torch.manual_seed(0)
spec = KernelSpec(kind, 2, [[1., 1.], [20., 20.]], jitter=1e-5)
u = spec.draw_initial_u().to(dev); m = spec.struct()
hist = torch.empty(T, spec.n_params, dtype=torch.float64, device=dev)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    _lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, T, _lib.ptr(hist), None))
    dt = time.time() - t
    print(f"fit N={N} T={T}: {dt/T*1e3:.3f} ms/iter, {N**3/(dt/T)/1e12:.2f} TFLOP/s (N^3 model)")
if M:
    g = np.stack(np.meshgrid(np.arange(int(np.sqrt(M))), np.arange(int(np.sqrt(M))), indexing="ij"), -1).reshape(-1, 2).astype(np.float64)
    Xs = torch.from_numpy(g).to(dev); M = len(g)
    mean = torch.empty(M, dtype=torch.float64, device=dev); var = torch.empty_like(mean)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        _lib.check(lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), _lib.ptr(Xs), M, _lib.ptr(mean), _lib.ptr(var)))
        dt = time.time() - t
        print(f"predict N={N} M={M}: {dt*1e3:.2f} ms, {(N*N*M + N**3*2/3)/dt/1e12:.2f} TFLOP/s")
if os.environ.get("PROF_STAGES"):
    lib.gpimhip_timing_enable(H.h, 1)
    _lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, T, _lib.ptr(hist), None))
    for st, nm in enumerate(["potrf", "trtri", "lauum"]):
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        lib.gpimhip_timing_read(H.h, st, ctypes.byref(ms), ctypes.byref(cnt))
        print("  stage %-6s %.3f ms per call (%d calls)" % (nm, ms.value / max(cnt.value, 1), cnt.value))
    lib.gpimhip_timing_enable(H.h, 0)
print("workspace GiB", lib.gpimhip_workspace_bytes(H.h) / 2**30)
H.close(); del H; torch.cuda.synchronize()
