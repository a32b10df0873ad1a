import sys, os, time, tempfile
import numpy as np, torch
R0 = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from problems import notebook_problem
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
    print("C4 %-40s %.3f s" % (tag, best), flush=True)
print("threads", torch.get_num_threads())
c4("default threads")
torch.set_num_threads(32)
a = torch.randn(2000, 2000, dtype=torch.float64); (a @ a).sum()
c4("after set_num_threads(32) + a CPU matmul")
torch.set_num_threads(1)
c4("set_num_threads(1)")
