"""Config C4 timed in a fresh process, then again after the concurrent C3 / C5 steps have run in the same process (their
host threads, streams and handles): does the process state slow the single-workgroup training kernel down?"""
import sys, os, time, tempfile, gc
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from gpim_amd import dist as gd
from problems import notebook_problem, hyperspectral_cube, ckpfm_cube
tmp = tempfile.mkdtemp()
def c4(tag):
    best = 1e9
    for rep in range(3):
        trial_func, Z = notebook_problem(4)
        bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), trial_func, acquisition_function="ei",
                             exploration_steps=30, verbose=0, filename=os.path.join(tmp, "bo"))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bo.run()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("C4 %-28s %.3f s" % (tag, best), flush=True)
c4("fresh process")
R, _ = hyperspectral_cube(size=64, nspec=64)
gd.reconstruct_slices(R, axis=-1, batch="auto", kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=50, verbose=0)
torch.cuda.synchronize()
c4("after C3 (16 x 4 threads)")
gc.collect(); torch.cuda.empty_cache()
c4("after gc + empty_cache")
gd.reconstruct_slices(ckpfm_cube(), axis=-1, sparse=True, indpoints=512, kernel="RBF", learning_rate=0.05, iterations=20)
torch.cuda.synchronize()
c4("after C5 (2 threads)")
