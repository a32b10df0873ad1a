"""C4: the time of every one of the 31 trainings of a BO run, for several runs (fast runs 0.285 s, slow ones 0.34-0.47 s)."""
import sys, os, time, tempfile
import numpy as np, torch
R0 = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from problems import notebook_problem
tmp = tempfile.mkdtemp()
for rep in range(8):
    trial_func, Z = notebook_problem(4)
    bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), trial_func, acquisition_function="ei",
                         exploration_steps=30, verbose=0, filename=os.path.join(tmp, "bo"))
    sm = bo.surrogate_model
    ts = []
    f = sm.train
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t)); return r
    sm.train = g
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bo.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("C4 %.3f s | " % dt + " ".join("%.1f" % v for v in ts), flush=True)
