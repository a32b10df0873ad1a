"""
Regenerates the fixtures under tests/golden/ that come from the reference checkout.

Run in the build container only (needs /root/reference; the GPU box never has it):
    python tests/tools/make_reference_fixtures.py

* test_{ei,poi,cb}.npy  -- the reference's own golden vectors (test/test_data/*.npy,
                           asserted at test/test_boptim.py:58), copied byte for byte (data).
* notebook_trace.json   -- the "Final parameter values" lines printed by the CPU run kept
                           in examples/notebooks/GP_based_exploration_exploitation.ipynb
                           (cell with boptim.run(), exploration_steps=50): one row per
                           1000-iteration training.
* host_logic.npz        -- outputs of the reference's importable NON-GP functions
                           (gprutils grid/prep helpers, acqfunc formulas, boptimizer
                           selection logic driven by a stub surrogate) on seeded inputs.
                           pyro / gpytorch are not installed, so they are replaced by
                           MagicMock for the import only; none of the recorded functions
                           touches them.
"""
import json
import os
import re
import shutil
import sys
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden")


def golden_npy():
    for a in ("ei", "poi", "cb"):
        shutil.copyfile(f"{REF}/test/test_data/test_{a}.npy", f"{OUT}/test_{a}.npy")


def notebook_trace():
    nb = json.load(open(f"{REF}/examples/notebooks/GP_based_exploration_exploitation.ipynb"))
    names = iter(["ei", "ei_mask", "ei_dscale", "custom"])   # the four boptim.run() cells, in order
    runs = {}
    for cell in nb["cells"]:
        if cell["cell_type"] != "code" or "exploration_steps=50" not in "".join(cell["source"]):
            continue
        text = "".join("".join(o.get("text", [])) for o in cell["outputs"]
                       if o.get("output_type") == "stream")
        rows = []
        for m in re.finditer(r"amp: ([\d.e+-]+), lengthscale: \[\s*([\d.e+-]+)\s+([\d.e+-]+)\s*\], noise: ([\d.e+-]+)", text):
            rows.append([float(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))])
        runs[next(names)] = rows
    meta = {"source": "examples/notebooks/GP_based_exploration_exploitation.ipynb",
            "setup": "np.random.seed(42); 25x25; idx=randint(0,25,(5,2)); 3-gaussian trial_func; "
                     "boptimizer(exploration_steps=50, use_gpu=False), defaults otherwise. "
                     "ei: acquisition 'ei'; ei_mask: + mask=mask_edges(ones,(2,2)); "
                     "ei_dscale: + dscale=4, memory=10; custom: acquisition mean + 5*sd",
            "columns": ["amp", "lengthscale0", "lengthscale1", "noise"],
            "printed_with": "np.around(amp,4), np.around(lengthscale,4), np.around(noise,7)",
            "runs": runs}
    json.dump(meta, open(f"{OUT}/notebook_trace.json", "w"))
    print("notebook rows:", {k: len(v) for k, v in runs.items()})


def host_logic():
    for name in ("pyro", "pyro.contrib", "pyro.contrib.gp", "pyro.distributions",
                 "pyro.infer", "gpytorch", "gpytorch.kernels", "gpytorch.constraints"):
        sys.modules[name] = mock.MagicMock()
    np.product = np.prod            # removed in NumPy 2
    sys.path.insert(0, REF)
    from gpim import gprutils
    from gpim.gpbayes import acqfunc
    from gpim.gpbayes.boptim import boptimizer

    out = {}
    rng = np.random.default_rng(7)
    # --- grids / prep ---
    R2 = rng.standard_normal((6, 9)); R2[rng.random((6, 9)) < 0.4] = np.nan
    R3a = rng.standard_normal((4, 5, 3)); R3a[rng.random((4, 5)) < 0.5] = np.nan      # xy sparsity
    R3b = rng.standard_normal((4, 5, 3)); R3b[rng.random((4, 5, 3)) < 0.5] = np.nan   # xyz sparsity
    R3b[0, 0, -1] = np.nan
    R4 = rng.standard_normal((3, 2, 4, 2))
    out["R2"], out["R3a"], out["R3b"], out["R4shape"] = R2, R3a, R3b, np.array(R4.shape)
    out["full2"] = gprutils.get_full_grid(R2)
    out["full2_dense"] = gprutils.get_full_grid(R2, dense_x=0.5)
    out["full2_extent"] = gprutils.get_full_grid(np.zeros((6, 8)), extent=[[2, 5], [1, 5]])
    out["full3"] = gprutils.get_full_grid(R3a)
    out["full4"] = gprutils.get_full_grid(R4)
    out["sparse2"] = gprutils.get_sparse_grid(R2)
    out["sparse3a"] = gprutils.get_sparse_grid(R3a)
    out["sparse3b"] = gprutils.get_sparse_grid(R3b)
    X, y = gprutils.prepare_training_data(out["sparse2"], R2)
    out["prep2_X"], out["prep2_y"] = X.numpy(), y.numpy()
    X, y = gprutils.prepare_training_data(out["sparse3a"], R3a)
    out["prep3a_X"], out["prep3a_y"] = X.numpy(), y.numpy()
    out["test2"] = gprutils.prepare_test_data(out["full2"]).numpy()
    out["test_sparse2"] = gprutils.prepare_test_data(out["sparse2"]).numpy()

    # --- acquisition formulas with a stub surrogate ---
    class Stub:
        def __init__(self, mean_full, sd_full, mean_obs, sd_obs):
            self.q = [(mean_full, sd_full), (mean_obs, sd_obs)]
            self.i = 0
        def predict(self, X, verbose=0):
            r = self.q[self.i % 2]; self.i += 1
            return r
    mf = rng.standard_normal((7, 5)); sf = rng.random((7, 5)) + 0.05
    mo = mf.copy(); so = sf.copy()
    holes = rng.random((7, 5)) < 0.7
    mo[holes] = np.nan; so[holes] = np.nan
    out["acq_mf"], out["acq_sf"], out["acq_mo"], out["acq_so"] = mf, sf, mo, so
    out["acq_cb"] = acqfunc.confidence_bound(Stub(mf, sf, mo, so), None, alpha=0.3, beta=1.7)[0]
    out["acq_ei"] = acqfunc.expected_improvement(Stub(mf, sf, mo, so), None, None, xi=0.01)[0]
    out["acq_poi"] = acqfunc.probability_of_improvement(Stub(mf, sf, mo, so), None, None, xi=0.01)[0]

    # --- selection logic (next_point ranking, checkvalues, update_points) ---
    class Surrogate:
        class model:
            class kernel:
                lengthscale = __import__("torch").tensor([2.0, 3.0], dtype=__import__("torch").float64)
        def __init__(self, m, s):
            self.m, self.s = m, s
        def predict(self, X, verbose=0):
            return self.m, self.s

    def make(acqf, **kw):
        Z = np.full((7, 5), np.nan); Z[1, 2] = 0.3; Z[4, 4] = -0.2
        bo = boptimizer.__new__(boptimizer)
        bo.verbose = 0
        bo.surrogate_model = Surrogate(mf, sf)
        bo.X_sparse = gprutils.get_sparse_grid(Z); bo.y_sparse = Z.copy()
        bo.X_full = gprutils.get_full_grid(Z)
        bo.acquisition_function = acqf
        bo.batch_update = kw.get("batch_update", False)
        bo.batch_size = kw.get("batch_size", 12)
        bo.alpha, bo.beta, bo.xi = 0.3, 1.7, 0.01
        bo.dscale = kw.get("dscale"); bo.batch_dscale = kw.get("batch_dscale")
        bo.batch_out_max = kw.get("batch_out_max", 4)
        bo.gamma, bo.points_mem = 0.8, kw.get("memory", 10)
        bo.exit_strategy = kw.get("exit_strategy", 0)
        bo.mask = kw.get("mask")
        bo.indices_all = kw.get("indices_all", [])
        bo.vals_all, bo.gp_predictions = [], []
        return bo

    v, i = make("cb").next_point()
    out["np_cb_vals"], out["np_cb_inds"] = np.array(v), np.array(i)
    mask = np.ones((7, 5)); mask[rng.random((7, 5)) < 0.3] = np.nan
    out["np_mask"] = mask
    v, i = make("cb", mask=mask).next_point()
    out["np_mask_vals"], out["np_mask_inds"] = np.array(v), np.array(i)
    hist = [list(map(int, out["np_cb_inds"][0])), list(map(int, out["np_cb_inds"][2]))]
    bo = make("cb", indices_all=[h[:] for h in hist])
    ind, val = bo.checkvalues(out["np_cb_inds"].tolist(), out["np_cb_vals"].tolist())
    out["cv_hist"], out["cv_ind"], out["cv_val"] = np.array(hist), np.array(ind), np.array(val)
    bo = make("cb", indices_all=[h[:] for h in hist], dscale=2.5)
    ind, val = bo.checkvalues(out["np_cb_inds"].tolist(), out["np_cb_vals"].tolist())
    out["cvd_ind"], out["cvd_val"] = np.array(ind), np.array(val)
    bo = make("cb", indices_all=out["np_cb_inds"].tolist(), exit_strategy=0)
    ind, val = bo.checkvalues(out["np_cb_inds"].tolist(), out["np_cb_vals"].tolist())
    out["cvx_ind"], out["cvx_val"] = np.array(ind), np.array(val)
    # batch update without random fill: large batch, small radius
    bo = make("cb", batch_update=True, batch_size=35, batch_dscale=1.5, batch_out_max=4)
    v, i = bo.next_point()
    out["bu_vals"], out["bu_inds"] = np.array(v), np.array(i)
    bo = make("cb", batch_update=True, batch_size=35, batch_out_max=3)   # kernel-lengthscale radius 2.5
    v, i = bo.next_point()
    out["bul_vals"], out["bul_inds"] = np.array(v), np.array(i)
    np.savez_compressed(f"{OUT}/host_logic.npz", **out)
    print("host logic keys:", len(out))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    golden_npy()
    notebook_trace()
    host_logic()
