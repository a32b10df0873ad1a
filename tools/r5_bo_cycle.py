"""Is a boptimizer (and its surrogate's library handle) released by reference counting alone?  Lists what the cyclic
collector finds otherwise."""
import sys, os, gc, weakref, tempfile
import numpy as np, torch
R0 = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from problems import notebook_problem
tmp = tempfile.mkdtemp()
trial_func, Z = notebook_problem(4)
gc.collect(); gc.disable()
bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), trial_func, acquisition_function="ei",
                     exploration_steps=3, verbose=0, filename=os.path.join(tmp, "bo"))
bo.run()
hw = weakref.ref(bo.surrogate_model._handle); bw = weakref.ref(bo); sw = weakref.ref(bo.surrogate_model)
del bo
print("after del: boptimizer alive %s, surrogate alive %s, handle alive %s" % (bw() is not None, sw() is not None, hw() is not None))
if hw() is not None:
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    print("collector found", n, "objects; types:", sorted({type(o).__name__ for o in gc.garbage})[:40])
    for o in gc.garbage:
        if type(o).__name__ in ("boptimizer", "reconstructor", "Handle", "_LazyMaps", "_Pending"):
            print("  ", type(o).__name__, "referrers:", [type(r).__name__ for r in gc.get_referrers(o)][:8])
