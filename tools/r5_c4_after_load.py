"""C4 right after seconds of chip-wide MFMA work (full C1 + C3), repetition by repetition: does the single-workgroup training
kernel run slower for a while (clocks)?"""
import sys, os, time, tempfile
import numpy as np, torch
R0 = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from gpim_amd import dist as gd
from problems import notebook_problem, hyperspectral_cube, spiral_pfm_image
import bench
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
    print("C4 %-30s %s" % (tag, " ".join("%.3f" % v for v in out)), flush=True)
c4("fresh", 4)
R = spiral_pfm_image()
X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
cube, _ = hyperspectral_cube()
for rnd in range(2):
    gpim.reconstructor(X, R, Xf, verbose=0, **bench.C1).run()
    gd.reconstruct_slices(cube, axis=-1, batch=16, batch_concurrency=4, **bench.C3)
    torch.cuda.synchronize()
    c4("after full C1 + C3", 4)
time.sleep(3)
c4("after 3 s idle", 3)
