"""sha1 of the factor gpimhip_potrf returns for a fixed SPD matrix (bit-level comparison of schedule variants across processes).
    python tools/r3_potrf_hash.py N"""
import ctypes, hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpim_amd import _lib
N = int(sys.argv[1]); dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
G = torch.randn(N, 64, generator=g, dtype=torch.float64)
A = (G @ G.T / 64 + torch.eye(N, dtype=torch.float64) * 2).to(dev).contiguous()
info = torch.zeros(1, dtype=torch.int32, device=dev)
H = _lib.Handle()
_lib.check(H.lib.gpimhip_potrf(H.h, _lib.ptr(A), N, N, _lib.ptr(info)))
torch.cuda.synchronize()
L = torch.tril(A).cpu().numpy()
print("N=%d info=%d sha1(L)=%s" % (N, int(info.item()), hashlib.sha1(L.tobytes()).hexdigest()[:16]))
