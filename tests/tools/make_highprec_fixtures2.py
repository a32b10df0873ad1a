"""
More independent 50-digit known answers (tests/golden/gp_highprec2.npz) for what VERDICT round 2 listed as pinned
by nothing external: an ISOTROPIC lengthscale (one shared parameter, d = 3) and a d = 4 exact GP (Matern52).
Same arithmetic as make_highprec_fixtures.py (imported: mpmath, dense N x N algebra, central differences at
1e-20), exact GP only.      python tests/tools/make_highprec_fixtures2.py
"""
import os

import mpmath as mp
import numpy as np

import make_highprec_fixtures as F

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "gp_highprec2.npz")


def expand(case, u):
    """isotropic cases carry ONE lengthscale parameter: replicate it for the d-dimensional routines"""
    if not case["iso"]:
        return u
    d = case["d"]
    return [u[0]] + [u[1]] * d + list(u[2:])


def loss(case, u):
    full = F.exact_loss(case, expand(case, u))
    if case["iso"]:       # the prior constant counts the single Uniform(lo, hi) once, not d times
        full -= (case["d"] - 1) * mp.log(case["ls_hi"][0] - case["ls_lo"][0])
    return full


def main():
    out = {}
    specs = [("Matern52", 4, 26, 6, False, 11), ("RBF", 3, 24, 6, True, 12), ("RationalQuadratic", 3, 22, 5, True, 13)]
    for ci, (kind, d, N, M, iso, seed) in enumerate(specs):
        case, pts, yv, Xs, u, _ = F.make_case(kind, d, N, M, 4, seed)
        case["iso"] = iso
        if iso:
            u = np.concatenate([u[:1], u[1:2], u[1 + d:]])
        umv = [mp.mpf(float(v)) for v in u]
        val = loss(case, umv)
        grad = [F.central(lambda v: loss(case, v), umv, k) for k in range(len(umv))]
        mean, var = F.exact_predict(case, expand(case, umv))
        tag = "c%d_" % ci
        out[tag + "kind"] = np.array(kind)
        out[tag + "iso"] = np.array(iso)
        out[tag + "X"], out[tag + "y"], out[tag + "Xs"], out[tag + "u"] = pts, yv, Xs, u
        out[tag + "ls"] = np.array([0.5, 9.0]) if iso else np.array([[0.5] * d, [9.0] * d])
        out[tag + "jitter"] = np.array(1e-5)
        out[tag + "loss"] = np.array(float(val))
        out[tag + "grad"] = np.array([float(g) for g in grad])
        out[tag + "mean"] = np.array([float(v) for v in mean])
        out[tag + "var"] = np.array([float(v) for v in var])
        print(kind, d, "iso" if iso else "ard", "loss %.15g" % float(val), flush=True)
    out["n_cases"] = np.array(len(specs))
    np.savez(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
