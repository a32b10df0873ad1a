"""Rank 0's compute per Adam iteration of the sharded symmetry-reduced model (gpim_amd.dist_symm) at world size 1 / 2 / 4 / 8
on one GPU, the all-reduce of eleven doubles replaced by a no-op (compute-only view; the loss printed is rank 0's share only).
usage: r5_symm_rank_share.py side(cube) [worlds...]"""
import sys, os, time, types
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from gpim_amd import dist_symm
side = int(sys.argv[1]); worlds = [int(w) for w in sys.argv[2:]] or [1, 2, 4, 8]
rng = np.random.default_rng(0)
g = np.meshgrid(*[np.arange(side, dtype=np.float64)] * 3, indexing="ij")
R = np.cos(g[0] / 9.0) * np.sin(g[1] / 7.0 + 0.3) * np.cos(g[2] / 11.0) + 0.05 * rng.standard_normal((side,) * 3)
X = gpim.utils.get_full_grid(R)
dist_symm.dist = types.SimpleNamespace(all_reduce=lambda *a, **k: None, is_available=lambda: False, is_initialized=lambda: False)
for world in worlds:
    dist_symm._world = lambda w=world: (0, w)
    kw = dict(kernel="Matern52", lengthscale=[[1.] * 3, [20.] * 3], learning_rate=0.1)
    dist_symm.symm_gp_fit(X, R, iterations=1, **kw)
    torch.cuda.synchronize(); t = time.perf_counter()
    dist_symm.symm_gp_fit(X, R, iterations=2, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2
    print("cube %d^3, world %d, rank 0 (%d of 8 blocks): %.3f s per Adam iteration (set-up included)" % (side, world, len(range(0, 8, world)), dt), flush=True)
    torch.cuda.empty_cache()
