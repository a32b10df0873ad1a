import sys, os, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from gpim_amd import dist as gd
from problems import ckpfm_cube
cube4 = ckpfm_cube()
kw5 = dict(kernel="RBF", learning_rate=0.05, iterations=200)
gd.reconstruct_slices(cube4[..., :1], axis=-1, sparse=True, indpoints=512, **dict(kw5, iterations=3))
ref = None
for conc in (1, 2, 4, 5):
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        m, s = gd.reconstruct_slices(cube4, axis=-1, sparse=True, indpoints=512, sparse_concurrency=conc, **kw5)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    if ref is None: ref = (m, s)
    print("C5 concurrency %d: %.3f s; identical to sequential: %s" % (conc, dt, np.array_equal(m, ref[0]) and np.array_equal(s, ref[1])), flush=True)
