"""Acquisition step of boptimizer on a large candidate grid (256 x 256, M = 65536), N ~ 60 observations:
the device-resident path (acquisition_on_device + radix top-k) against the round-1 path (public acqfunc
function: two full-grid predicts, maps copied to the host, single-workgroup top-k)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import gpim_amd as gpim
from gpim_amd import acqfunc

rng = np.random.default_rng(1)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ii, jj = np.meshgrid(np.arange(float(S)), np.arange(float(S)), indexing="ij")
truth = np.exp(-((ii - 0.23 * S) ** 2 + (jj - 0.7 * S) ** 2) / (3 * S)) + 0.6 * np.exp(-((ii - 0.74 * S) ** 2 + (jj - 0.27 * S) ** 2) / (6 * S))
Z = np.full((S, S), np.nan)
seed = rng.integers(0, S, size=(60, 2))
Z[seed[:, 0], seed[:, 1]] = truth[seed[:, 0], seed[:, 1]]
bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), lambda idx: truth[tuple(idx)],
                     acquisition_function="ei", exploration_steps=1, gp_iterations=200, verbose=0,
                     lengthscale=[[1., 1.], [60., 60.]])
bo.surrogate_model.train()
sync = torch.cuda.synchronize

def new_path():
    return bo.next_point()

def old_path():
    acq, pred = acqfunc.expected_improvement(bo.surrogate_model, bo.X_full, bo.X_sparse, xi=bo.xi)
    bo.gp_predictions.append(pred)
    return bo._rank(np.asarray(acq))

for name, fn, env in (("device-resident (round 2)", new_path, None), ("round-1 path", old_path, "1")):
    if env: os.environ["GPIMHIP_NO_RADIX_TOPK"] = env
    for _ in range(6):                       # warm-up incl. torch's caching allocator (retained maps)
        fn()
    sync()
    t = time.perf_counter()
    for _ in range(5):
        v, i = fn()
    sync()
    print("%-28s %.2f ms per acquisition step (M = %d, N = %d); first pick %s" % (name, (time.perf_counter() - t) / 5 * 1e3, S * S, 60, i[0]))

if os.environ.get("BO_PROFILE"):
    import cProfile, pstats
    os.environ.pop("GPIMHIP_NO_RADIX_TOPK", None)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        new_path()
    sync()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
