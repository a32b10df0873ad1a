"""Config C4 (BO 25x25, EI, 30 steps x 1000 Adam iterations): where a step's time goes -- training, acquisition sweep + ranking,
host logic (checkvalues, target evaluation, data preparation), saving."""
import sys, os, time, tempfile
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from problems import notebook_problem
tmp = tempfile.mkdtemp()
for rep in range(8):
    trial_func, Z = notebook_problem(4)
    bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), trial_func, acquisition_function="ei",
                         exploration_steps=30, verbose=0, filename=os.path.join(tmp, "bo"))
    T = {"train": 0.0, "next_point": 0.0, "checkvalues": 0.0, "evaluate": 0.0, "prepare": 0.0, "save": 0.0}
    sm = bo.surrogate_model
    def timed(name, f):
        def g(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); T[name] += time.perf_counter() - t; return r
        return g
    sm.train = timed("train", sm.train)
    bo.next_point = timed("next_point", bo.next_point)
    bo.checkvalues = timed("checkvalues", bo.checkvalues)
    bo.evaluate_function = timed("evaluate", bo.evaluate_function)
    bo.save_results = timed("save", bo.save_results)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bo.run()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("C4 %.3f s | " % dt + " ".join("%s %.1f" % (k, 1e3 * v) for k, v in T.items()) + " other %.1f ms" % (1e3 * (dt - sum(T.values()))), flush=True)
