"""
Full replay of the reference notebook's four BO runs on the CPU oracle (51 trainings of 1000 Adam
iterations each, ~10 minutes in total on one core): how many of the printed hyper-parameter rows of
examples/notebooks/GP_based_exploration_exploitation.ipynb (tests/golden/notebook_trace.json) the
oracle reproduces to every printed digit.  The CPU suite (tests/test_oracle_golden.py) replays only
the first rows of each run; this script is the evidence for the full-trace numbers in DESIGN.md.

    python tests/tools/pin_oracle_full.py [run ...]      # runs: ei ei_mask ei_dscale custom

Writes tests/golden/oracle_full_replay.json: per run the number of exactly matching rows, the indices
of the rows that differ and by how much, and whether the queried points coincide afterwards.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_oracle_golden import oracle_factory, run_notebook     # noqa: E402

TOL = np.array([1.01e-4, 1.01e-4, 1.01e-4, 1.01e-7])     # printed with 4 / 4 / 4 / 7 decimals


def main(which_all):
    import torch
    torch.set_num_threads(1)
    trace = json.load(open(os.path.join(ROOT, "tests", "golden", "notebook_trace.json")))["runs"]
    out_path = os.path.join(ROOT, "tests", "golden", "oracle_full_replay.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for which in which_all:
        expected = np.array(trace[which])
        rows, bo = run_notebook(which, len(expected) - 1, oracle_factory)
        exact = (np.abs(rows - expected) <= TOL).all(axis=1)
        bad = np.flatnonzero(~exact)
        out[which] = {
            "rows": int(len(rows)), "exact_rows": int(exact.sum()),
            "differing_rows": [{"row": int(r), "oracle": rows[r].tolist(), "notebook": expected[r].tolist()}
                               for r in bad],
            "max_rel_diff_first3": float(np.max(np.abs(rows[:, :3] - expected[:, :3]) /
                                                np.maximum(np.abs(expected[:, :3]), 1e-12))),
            "last_row_exact": bool(exact[-1]),
        }
        print(which, "%d/%d rows exact; differing rows: %s" % (exact.sum(), len(rows), bad.tolist()), flush=True)
        json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:] or ["ei", "ei_mask", "ei_dscale", "custom"])
