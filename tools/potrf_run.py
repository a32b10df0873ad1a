"""Runs gpimhip_potrf alone (SPD matrix = kernel matrix + noise) REPS times at each size; used under
rocprofv3 --kernel-trace by tools/potrf_timeline.py.   usage: potrf_run.py N [N ...]"""
import ctypes, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpim_amd import _lib
dev = torch.device("cuda:0")
H = _lib.Handle(); lib = H.lib
for N in [int(a) for a in sys.argv[1:]]:
    g = torch.Generator(device=dev); g.manual_seed(0)
    x = torch.rand(N, 2, dtype=torch.float64, device=dev, generator=g) * 100
    K = torch.exp(-0.5 * torch.cdist(x, x) ** 2 / 25.0)
    K.diagonal().add_(0.1)
    info = torch.zeros(2, dtype=torch.int32, device=dev)
    A = torch.empty_like(K)
    ts = []
    for rep in range(4):
        A.copy_(K)
        torch.cuda.synchronize(); t = time.perf_counter()
        _lib.check(lib.gpimhip_potrf(H.h, _lib.ptr(A), N, N, _lib.ptr(info)))
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    L = torch.tril(A); v = torch.randn(N, 1, dtype=torch.float64, device=dev)
    res = float((L @ (L.T @ v) - K @ v).norm() / (K @ v).norm())
    import hashlib
    print("   residual |L L^T v - K v| / |K v| = %.2e   sha1(L) %s" % (res, hashlib.sha1(L.cpu().numpy().tobytes()).hexdigest()[:12]))
    print("potrf N=%d: %s ms  -> %.1f TFLOP/s  info %d" % (N, ["%.2f" % v for v in ts], N ** 3 / 3 / min(ts) / 1e9, int(info[0])), flush=True)
    del K, A
