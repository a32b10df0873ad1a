"""Which earlier work in the same process slows the look-ahead factorisation?  usage: c2s_probe.py [c1|c3|c4|c5|kron|none]"""
import sys, time, os, gc; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
import gpim_amd as gpim
from gpim_amd import dist as gdist
from problems import lattice_image, spiral_image, hyperspectral_cube, ckpfm_cube, notebook_problem
sync = torch.cuda.synchronize
pres = sys.argv[1:] if len(sys.argv) > 1 else ["none"]
for pre in pres:
  if pre == "c1":
      R, _ = spiral_image(); X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
      gpim.reconstructor(X, R, Xf, kernel="RBF", lengthscale=[[1., 1.], [4., 4.]], learning_rate=0.1, iterations=20, verbose=0).run()
  elif pre == "c3":
      cube, _ = hyperspectral_cube()
      gdist.reconstruct_slices(cube, axis=-1, batch=64, kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=10)
  elif pre == "c4":
      f, Z = notebook_problem(4)
      gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), f, acquisition_function="ei", exploration_steps=3, verbose=0, filename="/tmp/bo").run()
  elif pre == "c5":
      cube4 = ckpfm_cube()
      gdist.reconstruct_slices(cube4[..., :1], axis=-1, sparse=True, indpoints=512, kernel="RBF", learning_rate=0.05, iterations=5)
  elif pre == "kron":
      cube4 = ckpfm_cube(); R5 = cube4[..., 0]; Xf5 = gpim.utils.get_full_grid(R5)
      gpim.reconstructor(Xf5, R5, Xf5, structured=True, verbose=0, kernel="RBF", learning_rate=0.05, iterations=5).run()
gc.collect(); sync()
R2, _ = lattice_image(size=256, frac=0.25, seed=1)
X2, Xf2 = gpim.utils.get_sparse_grid(R2), gpim.utils.get_full_grid(R2)
for prec in ("single", "double"):
    rec2 = gpim.reconstructor(X2, R2, Xf2, iterations=2, kernel="Matern52", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, verbose=0, seed=0, precision=prec)
    rec2.run()
    rec2.iterations = 20
    sync(); t0 = time.perf_counter()
    rec2.train()
    sync(); t1 = time.perf_counter()
    print("after %-14s %s: %.2f ms per iteration" % ("+".join(pres), prec, (t1 - t0) / 20 * 1e3), flush=True)
    del rec2; gc.collect()
