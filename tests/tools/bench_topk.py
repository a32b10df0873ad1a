import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gpim_amd import _lib
H = _lib.Handle()
rng = np.random.default_rng(0)
for M in (65536, 1 << 20):
    x = torch.from_numpy(rng.standard_normal(M)).cuda()
    k = 100
    vals = torch.empty(k, dtype=torch.float64, device="cuda"); idx = torch.empty(k, dtype=torch.int64, device="cuda"); cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
    for env in (None, "1"):
        if env: os.environ["GPIMHIP_NO_RADIX_TOPK"] = "1"
        else: os.environ.pop("GPIMHIP_NO_RADIX_TOPK", None)
        for rep in range(2):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(10):
                _lib.check(H.lib.gpimhip_topk(H.h, _lib.ptr(x), M, k, 1, _lib.ptr(vals), _lib.ptr(idx), _lib.ptr(cnt)))
            torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
        print("M=%d %s: %.3f ms" % (M, "old" if env else "radix", dt * 1e3))
    out = torch.empty(1, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): _lib.check(H.lib.gpimhip_nanmax(H.h, _lib.ptr(x), M, _lib.ptr(out)))
    torch.cuda.synchronize(); print("nanmax two-stage %.3f ms" % ((time.perf_counter() - t) / 10 * 1e3))
