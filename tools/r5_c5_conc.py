"""Config C5 on one GPU (5 sparse slices, T = 200): lock-step batch size x concurrent host threads (sparse_batch 0 = one
reconstructor per slice)."""
import sys, os, time
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
from gpim_amd import dist as gd
from problems import ckpfm_cube
cube4 = ckpfm_cube()
kw5 = dict(kernel="RBF", learning_rate=0.05, iterations=200)
gd.reconstruct_slices(cube4[..., :1], axis=-1, sparse=True, indpoints=512, **dict(kw5, iterations=3))
for sb, conc in ((0, 1), (0, 4), (5, 1), (3, 1), (3, 2), (2, 2), (2, 3)):
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        gd.reconstruct_slices(cube4, axis=-1, sparse=True, indpoints=512, sparse_batch=sb, sparse_concurrency=conc, **kw5)
        torch.cuda.synchronize(); best = min(best, time.time() - t)
    print("C5, sparse_batch %d, sparse_concurrency %d: %.3f s" % (sb, conc, best), flush=True)
