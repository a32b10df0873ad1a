"""A whole complete cube as ONE exact GP through the reflection blocks (8 blocks of N / 8 points): seconds per Adam
iteration and workspace.   usage: r5_symm_cube.py side [T]"""
import sys, os, time
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
side = int(sys.argv[1]); T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.default_rng(0)
g = np.meshgrid(*[np.arange(side, dtype=np.float64)] * 3, indexing="ij")
R = np.cos(g[0] / 9.0) * np.sin(g[1] / 7.0 + 0.3) * np.cos(g[2] / 11.0) + 0.05 * rng.standard_normal((side,) * 3)
X = gpim.utils.get_full_grid(R)
kw = dict(kernel="Matern52", lengthscale=[[1.] * 3, [20.] * 3], learning_rate=0.1, verbose=0)
rec = gpim.reconstructor(X, R, X, structured=True, iterations=1, **kw)
rec.train()
torch.cuda.synchronize(); t = time.perf_counter()
rec.train(iterations=T)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / T
N = side ** 3
print("cube %d^3 (N = %d, 8 blocks of %d): %.3f s per Adam iteration (dense-equivalent %.0f TFLOP/s over N^3; the blocks' own flop %.1f TFLOP/s); loss %s"
      % (side, N, N // 8, dt, float(N) ** 3 / dt / 1e12, 8 * float(N // 8) ** 3 / dt / 1e12, np.round(rec.loss_all, 3)))
print("workspace GiB %.1f" % (rec._handle.lib.gpimhip_workspace_bytes(rec._handle.h) / 2 ** 30))
torch.cuda.synchronize(); t = time.perf_counter()
mean, sd = rec.predict()
torch.cuda.synchronize(); dp = time.perf_counter() - t
print("prediction on the whole grid (M = %d): %.2f s; rmse(mean - data) %.4f, median sd %.4f; workspace GiB %.1f"
      % (N, dp, float(np.sqrt(np.mean((mean - R) ** 2))), float(np.median(sd)), rec._handle.lib.gpimhip_workspace_bytes(rec._handle.h) / 2 ** 30))
