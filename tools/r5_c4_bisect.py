"""Which of the bench extras that run before C4 slows it down (0.285 s in a fresh process, 0.38-0.47 s inside the extras child)?
usage: r5_c4_bisect.py [c1] [c3] [oracle] ... (the steps to run before C4, in order)"""
import sys, os, time, tempfile
import numpy as np, torch
R0 = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tests"))
import gpim_amd as gpim
from gpim_amd import dist as gd
from problems import notebook_problem, hyperspectral_cube, spiral_pfm_image
import bench
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
c4("fresh")
for step in sys.argv[1:]:
    if step == "c1":
        R = spiral_pfm_image()
        X, Xf = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
        gpim.reconstructor(X, R, Xf, **dict(bench.C1, iterations=20, verbose=0)).run()
    elif step == "c3":
        cube, _ = hyperspectral_cube()
        gd.reconstruct_slices(cube, axis=-1, batch=16, batch_concurrency=4, **dict(bench.C3, iterations=20))
    elif step == "c3r8":
        cube, _ = hyperspectral_cube()
        gd.reconstruct_slices(cube[..., 0::8], axis=-1, batch="auto", **dict(bench.C3, iterations=20))
    elif step == "oracle":
        cube, _ = hyperspectral_cube()
        bench.rmse_vs_oracle_c3(cube)
    torch.cuda.synchronize()
    c4("after " + step)
