"""C4 (README.md:71-106 instance) on the HIP engine against the oracle: index sequence, and how the 31 x 1000 hyper-parameter
rows compare (all rows, rows within 1e-7, the end-of-training rows)."""
import os, sys, tempfile
import numpy as np, torch
R0 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from oracle import gpim_oracle as O
from problems import notebook_problem
f, Z = notebook_problem(4)
tmp = tempfile.mkdtemp()
bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z.copy(), gpim.utils.get_full_grid(Z), f, acquisition_function="ei",
                     exploration_steps=30, verbose=0, filename=os.path.join(tmp, "a"))
bo.run()
torch.set_num_threads(8)
ob = O.boptimizer(O.get_sparse_grid(Z), Z.copy(), O.get_full_grid(Z), f, acquisition_function="ei", exploration_steps=30,
                  verbose=0, filename=os.path.join(tmp, "b"))
ob.run()
print("indices equal:", [tuple(int(v) for v in i) for i in bo.indices_all] == [tuple(int(v) for v in i) for i in ob.indices_all])
hh, ho = bo.surrogate_model.hyperparams, ob.surrogate_model.hyperparams
for key in ("variance", "lengthscale", "noise"):
    a, b = np.asarray(hh[key], float).reshape(31000, -1), np.asarray(ho[key], float).reshape(31000, -1)
    rel = np.abs(a - b) / np.abs(b)
    per_train = rel.reshape(31, 1000, -1).max(axis=(1, 2))
    ends = rel.reshape(31, 1000, -1)[:, -1].max(axis=1)
    print(key, "max rel %.3e; rows within 1e-7: %.4f; trainings with any row > 1e-7: %s" % (rel.max(), (rel.max(axis=1) <= 1e-7).mean(), np.nonzero(per_train > 1e-7)[0].tolist()))
    print("   per-training max rel:", " ".join("%.1e" % v for v in per_train))
    print("   end-of-training rel :", " ".join("%.1e" % v for v in ends))
