"""Per-launch content of the step plan (host side, no GPU): k-blocks of trailing update and of the inverse (T / X phases) per
launch, number of tile operations, deepest.   usage: r5_plan_stats.py nb"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpim_amd import _lib
lib = _lib.load()
nb = int(sys.argv[1])
n = ctypes.c_int64()
assert lib.gpimhip_step_plan_host(nb, 1, None, 0, ctypes.byref(n)) == 0
buf = np.zeros((n.value, 6), dtype=np.int32)
assert lib.gpimhip_step_plan_host(nb, 1, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n.value, ctypes.byref(n)) == 0
L, ci, cj, k0, k1, kind = buf.T
d = k1 - k0
tot_inv = d[kind != 0].sum(); tot_upd = d[kind == 0].sum()
print("nb %d: update k-blocks %d, inverse k-blocks %d (hosted in step launches: %d = %.0f %%)" % (nb, tot_upd, tot_inv, d[(kind != 0) & (L < nb)].sum(), 100.0 * d[(kind != 0) & (L < nb)].sum() / tot_inv))
print("launch: ops  upd_kb  inv_kb  maxdepth | cumulative inverse fraction")
cum = 0
for l in range(L.max() + 1):
    m = L == l
    cum += d[m & (kind != 0)].sum()
    if l % 4 == 0 or l >= nb:
        print("%4d: %5d %7d %7d %5d | %.3f" % (l, m.sum(), d[m & (kind == 0)].sum(), d[m & (kind != 0)].sum(), d[m].max() if m.any() else 0, cum / tot_inv))
