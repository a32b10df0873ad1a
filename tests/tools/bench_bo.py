"""Times the C4 configuration (BASELINE.json configs[0]): Bayesian optimisation on the 25x25 test
problem, EI, `steps` exploration steps x 1000 Adam iterations each.  usage: bench_bo.py [steps] [acq]"""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import gpim_amd as gpim
from problems import bo_test_problem

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
acq = sys.argv[2] if len(sys.argv) > 2 else "ei"
tmp = tempfile.mkdtemp()
for rep in range(3):
    trial_func, Z_sparse = bo_test_problem()
    bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z_sparse), Z_sparse, gpim.utils.get_full_grid(Z_sparse),
                         trial_func, acquisition_function=acq, exploration_steps=steps, verbose=0,
                         filename=os.path.join(tmp, "bo"))
    torch.cuda.synchronize(); t = time.time()
    bo.run()
    torch.cuda.synchronize(); dt = time.time() - t
    nfit = steps + 1
    print("BO %s, %d steps (%d trainings x 1000 its): %.3f s total, %.1f us per Adam iteration incl. acquisition"
          % (acq, steps, nfit, dt, dt / (nfit * 1000) * 1e6))
