"""Phase times of the distributed exact GP on one GPU (P = 1): K build, factorisation, solves, variance.
usage: dist_time.py [N]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpim_amd.dist_chol import DistributedCholesky, NB
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
side = int(np.sqrt(N))
ii, jj = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64), indexing="ij")
X = torch.from_numpy(np.stack([ii.ravel(), jj.ravel()], 1)).to(dev)
def cols(c0, c1):
    d = torch.cdist(X, X[c0:c1])
    K = 0.05 * (1 + np.sqrt(5) * d / 4 + 5 * d * d / 48) * torch.exp(-np.sqrt(5) * d / 4)
    idx = torch.arange(c0, c1, device=dev)
    K[idx, idx - c0] += 4e-4 + 1e-5
    return K
for rep in range(2):
    ch = DistributedCholesky(N)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ch.set_from_function(cols)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ch.factor()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    m = 8192
    B = torch.randn((ch.layout.np, m), dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    q = ch.solve_colsumsq(B)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("N=%d: build %.3f s | factor %.3f s = %.1f TFLOP/s | colsumsq m=%d %.3f s = %.1f TFLOP/s" % (
        N, t1 - t0, t2 - t1, N ** 3 / 3 / (t2 - t1) / 1e12, m, t4 - t3, float(N) ** 2 * m / (t4 - t3) / 1e12), flush=True)
    del ch, B
