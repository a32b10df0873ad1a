"""Triage driver for the rocprofv3 crash of `bench.py --workload c3` (round 3): runs the C3 slices through
gpim_amd.dist.reconstruct_slices with a chosen lock-step batch / number of concurrent batches, several times in one
process, and checks every repetition bitwise against the first.  Writes /proc/self/maps next to its log so that the
frames of a crash report can be resolved against the libraries of that very process.

    python tools/r3_c3_triage.py <tag> <batch> <batch_concurrency> <reps> [iterations]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gpim_amd import _lib, dist as gdist          # noqa: E402
from problems import hyperspectral_cube           # noqa: E402

tag, batch, conc, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 250
out = os.path.join(ROOT, "gpurun_out", "c3tri")
os.makedirs(out, exist_ok=True)
C3 = dict(kernel="RBF", lengthscale=[[1., 1.], [20., 20.]], learning_rate=0.1, iterations=iters)
cube, _ = hyperspectral_cube()
H = _lib.Handle()                                  # loads libgpimhip.so and initialises the device
with open(os.path.join(out, tag + ".maps"), "w") as f:
    f.write(open("/proc/self/maps").read())
first = None
for r in range(reps):
    t0 = time.perf_counter()
    if conc > 1:
        mean, sd = gdist.reconstruct_slices(cube, axis=-1, batch=batch, batch_concurrency=conc, **C3)
    else:
        mean, sd = gdist.reconstruct_slices(cube, axis=-1, batch=batch, handle=H, **C3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(np.isfinite(mean).all() and np.isfinite(sd).all())
    same = None
    if first is None:
        first = (mean.copy(), sd.copy())
    else:
        same = bool(np.array_equal(first[0], mean) and np.array_equal(first[1], sd))
    print("%s rep %d: %.3f s finite=%s same_as_first=%s" % (tag, r, dt, ok, same), flush=True)
np.save(os.path.join(out, tag + "_mean.npy"), first[0].astype(np.float64))
print(tag, "done", flush=True)
