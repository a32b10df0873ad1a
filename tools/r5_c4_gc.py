"""C4 repetition by repetition with the cyclic collector on / off / frozen: are the sporadic +40-60 ms full collections?"""
import sys, os, time, tempfile, gc
import numpy as np, torch
R0 = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from problems import notebook_problem
tmp = tempfile.mkdtemp()
def c4(tag, reps):
    out = []
    for rep in range(reps):
        trial_func, Z = notebook_problem(4)
        bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), trial_func, acquisition_function="ei",
                             exploration_steps=30, verbose=0, filename=os.path.join(tmp, "bo"))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bo.run()
        torch.cuda.synchronize(); out.append(time.perf_counter() - t0)
    print("C4 %-22s %s" % (tag, " ".join("%.3f" % v for v in out)), flush=True)
c4("warm-up", 2)
def cb(phase, info):
    if phase == "start": cb.t = time.perf_counter()
    else: print("   gc gen %d: %.1f ms, collected %d" % (info["generation"], 1e3 * (time.perf_counter() - cb.t), info["collected"]), flush=True)
gc.callbacks.append(cb)
c4("collector on", 8)
gc.callbacks.remove(cb)
gc.disable()
c4("collector off", 8)
gc.enable(); gc.collect(); gc.freeze()
c4("collector on, frozen", 8)
