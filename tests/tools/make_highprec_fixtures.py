"""
Independent high-precision known answers for the models the reference has no golden vectors for
(Matern52 / RationalQuadratic exact GPs and the sparse VFE model): tests/golden/gp_highprec.npz.

The oracle (oracle/gpim_oracle.py) and the HIP engine are restatements by the same author; this script
is a THIRD evaluation that shares no code and no algebra with them: 50-digit mpmath arithmetic on the
published formulas, dense N x N linear algebra only (no Woodbury identity, no Cholesky-of-capacitance
form, no autograd) and central finite differences at step 1e-20 for every gradient component.

  exact GP   loss(u) = 1/2 y^T Kt^-1 y + 1/2 log det Kt + N/2 log 2 pi + log(prior widths),
             Kt = K + (jitter + noise) I                              (Rasmussen & Williams eq. 2.30;
             MAP objective of pyro.contrib.gp.models.GPRegression with Uniform priors)
             posterior  mean = K*^T Kt^-1 y,  var = max(k** - diag(K*^T Kt^-1 K*), 0) + noise   (eq. 2.25/2.26)
  sparse VFE loss(u, Xu) = -log N(y | 0, Qff + noise I) + tr(Kff - Qff) / (2 noise) + log(prior widths),
             Qff = Kfu (Kuu + jitter I)^-1 Kuf                        (Titsias 2009, eq. 9;
             pyro.contrib.gp.models.SparseGPRegression(approx="VFE"))
             posterior  S = (Kuu' + Kuf Kfu / noise)^-1,  mean = K*u S Kuf y / noise,
                        var = k** - K*u Kuu'^-1 Ku* + K*u S Ku* + noise              (Titsias 2009, eq. 6)
  kernels    RBF s2 exp(-r2/2);  Matern52 s2 (1 + sqrt5 r + 5/3 r2) exp(-sqrt5 r), r = sqrt(r2 + 1e-12) as Pyro
             (pyro-ppl 1.x isotropic.py: the shift enters through r only, the 5/3 term takes the un-shifted r2);
             RationalQuadratic s2 (1 + r2 / (2 alpha))^-alpha;   r2 = sum_k ((x_k - z_k) / l_k)^2
  parameters variance = lo + (hi - lo) sigmoid(u_0), lengthscale_k likewise, noise = exp(u), alpha = exp(u)
             (torch.distributions transform_to(interval / positive))

Run in the build container (mpmath is installed there; the GPU box only reads the .npz):
    python tests/tools/make_highprec_fixtures.py
"""
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 50
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "gp_highprec.npz")
AMP = (mp.mpf("1e-4"), mp.mpf(10))
H = mp.mpf("1e-20")


def sigmoid(u):
    return 1 / (1 + mp.e ** (-u))


def theta(kind, d, ls_lo, ls_hi, u):
    var = AMP[0] + (AMP[1] - AMP[0]) * sigmoid(u[0])
    ls = [ls_lo[k] + (ls_hi[k] - ls_lo[k]) * sigmoid(u[1 + k]) for k in range(d)]
    noise = mp.e ** u[1 + d]
    alpha = mp.e ** u[2 + d] if kind == "RationalQuadratic" else None
    return var, ls, noise, alpha


def kfun(kind, var, ls, alpha, x, z):
    r2 = sum(((x[k] - z[k]) / ls[k]) ** 2 for k in range(len(ls)))
    if kind == "RBF":
        return var * mp.e ** (-r2 / 2)
    if kind == "Matern52":
        r = mp.sqrt(r2 + mp.mpf("1e-12"))
        s5r = mp.sqrt(5) * r
        return var * (1 + s5r + mp.mpf(5) / 3 * r2) * mp.e ** (-s5r)
    return var * (1 + r2 / (2 * alpha)) ** (-alpha)


def gram(kind, var, ls, alpha, A, B):
    return mp.matrix([[kfun(kind, var, ls, alpha, a, b) for b in B] for a in A])


def solve(A, B):
    """A^-1 B through the explicit 50-digit inverse (condition numbers here are < 1e8)."""
    return mp.inverse(A) * B


def logdet_solve(Kt, y):
    L = mp.cholesky(Kt)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(Kt.rows))
    return logdet, solve(Kt, y)


def prior_const(d, ls_lo, ls_hi):
    return mp.log(AMP[1] - AMP[0]) + sum(mp.log(ls_hi[k] - ls_lo[k]) for k in range(d))


def exact_loss(case, u):
    kind, d, X, y, jitter = case["kind"], case["d"], case["X"], case["y"], case["jitter"]
    var, ls, noise, alpha = theta(kind, d, case["ls_lo"], case["ls_hi"], u)
    N = len(X)
    Kt = gram(kind, var, ls, alpha, X, X) + (jitter + noise) * mp.eye(N)
    logdet, a = logdet_solve(Kt, y)
    quad = sum(y[i] * a[i] for i in range(N))
    return quad / 2 + logdet / 2 + mp.mpf(N) / 2 * mp.log(2 * mp.pi) + prior_const(d, case["ls_lo"], case["ls_hi"])


def exact_predict(case, u):
    kind, d, X, y, jitter, Xs = case["kind"], case["d"], case["X"], case["y"], case["jitter"], case["Xs"]
    var, ls, noise, alpha = theta(kind, d, case["ls_lo"], case["ls_hi"], u)
    N = len(X)
    Kt = gram(kind, var, ls, alpha, X, X) + (jitter + noise) * mp.eye(N)
    Ks = gram(kind, var, ls, alpha, X, Xs)                # N x M
    a = solve(Kt, y)
    B = solve(Kt, Ks)
    mean = [sum(Ks[i, j] * a[i] for i in range(N)) for j in range(len(Xs))]
    v = [var - sum(Ks[i, j] * B[i, j] for i in range(N)) for j in range(len(Xs))]
    return mean, [max(t, mp.mpf(0)) + noise for t in v]


def vfe_loss(case, u, Xu):
    kind, d, X, y, jitter = case["kind"], case["d"], case["X"], case["y"], case["jitter"]
    var, ls, noise, alpha = theta(kind, d, case["ls_lo"], case["ls_hi"], u)
    N, M = len(X), len(Xu)
    Kuu = gram(kind, var, ls, alpha, Xu, Xu) + jitter * mp.eye(M)
    Kuf = gram(kind, var, ls, alpha, Xu, X)
    Qff = Kuf.T * solve(Kuu, Kuf)
    logdet, a = logdet_solve(Qff + noise * mp.eye(N), y)
    quad = sum(y[i] * a[i] for i in range(N))
    trace = sum(var - Qff[i, i] for i in range(N)) / noise
    trace = max(trace, mp.mpf(0))
    return (quad / 2 + logdet / 2 + mp.mpf(N) / 2 * mp.log(2 * mp.pi) + trace / 2 +
            prior_const(d, case["ls_lo"], case["ls_hi"]))


def vfe_predict(case, u, Xu):
    kind, d, X, y, jitter, Xs = case["kind"], case["d"], case["X"], case["y"], case["jitter"], case["Xs"]
    var, ls, noise, alpha = theta(kind, d, case["ls_lo"], case["ls_hi"], u)
    M = len(Xu)
    Kuu = gram(kind, var, ls, alpha, Xu, Xu) + jitter * mp.eye(M)
    Kuf = gram(kind, var, ls, alpha, Xu, X)
    Kus = gram(kind, var, ls, alpha, Xu, Xs)
    Sinv = Kuu + Kuf * Kuf.T / noise
    rhs = Kuf * y / noise
    mean_w = solve(Sinv, rhs)
    mean = [sum(Kus[i, j] * mean_w[i] for i in range(M)) for j in range(len(Xs))]
    A = solve(Kuu, Kus)
    B = solve(Sinv, Kus)
    v = [var - sum(Kus[i, j] * A[i, j] for i in range(M)) + sum(Kus[i, j] * B[i, j] for i in range(M)) + noise
         for j in range(len(Xs))]
    return mean, v


def central(f, vec, idx):
    vp, vm = list(vec), list(vec)
    vp[idx] += H
    vm[idx] -= H
    return (f(vp) - f(vm)) / (2 * H)


def make_case(kind, d, N, M, nu, seed):
    rng = np.random.default_rng(seed)
    pts = np.unique(rng.integers(0, 12, size=(8 * N, d)), axis=0)
    pts = pts[rng.permutation(len(pts))[:N]].astype(np.float64)
    yv = np.sin(pts.sum(1) / 3.0) + 0.1 * rng.standard_normal(N)
    Xs = rng.uniform(0, 11, size=(M, d))
    P = 2 + d + (1 if kind == "RationalQuadratic" else 0)
    u = rng.uniform(-1.0, 1.0, size=P)
    u[1 + d] = -2.0 + 0.3 * rng.standard_normal()          # noise ~ 0.1
    Xu = pts[:: max(1, N // nu)][:nu] + 0.25                # inducing inputs off the data points
    case = dict(kind=kind, d=d, jitter=mp.mpf("1e-5"),
                ls_lo=[mp.mpf("0.5")] * d, ls_hi=[mp.mpf(9)] * d,
                X=[[mp.mpf(float(v)) for v in r] for r in pts], y=mp.matrix([mp.mpf(float(v)) for v in yv]),
                Xs=[[mp.mpf(float(v)) for v in r] for r in Xs])
    return case, pts, yv, Xs, u, Xu


def main():
    out = {}
    specs = [("RBF", 2, 24, 9, 5, 1), ("Matern52", 2, 28, 9, 6, 2), ("RationalQuadratic", 2, 24, 9, 5, 3),
             ("Matern52", 3, 30, 7, 6, 4)]
    for ci, (kind, d, N, M, nu, seed) in enumerate(specs):
        case, pts, yv, Xs, u, Xu = make_case(kind, d, N, M, nu, seed)
        umv = [mp.mpf(float(v)) for v in u]
        P = len(u)
        loss = exact_loss(case, umv)
        grad = [central(lambda v: exact_loss(case, v), umv, k) for k in range(P)]
        mean, var = exact_predict(case, umv)
        Xuv = [mp.mpf(float(v)) for v in Xu.reshape(-1)]
        nxu = len(Xuv)

        def vl(vec):
            uu, xx = vec[:P], vec[P:]
            return vfe_loss(case, uu, [xx[i * d:(i + 1) * d] for i in range(len(Xu))])
        full = umv + Xuv
        vloss = vl(full)
        vgrad = [central(vl, full, k) for k in range(P + nxu)]
        vmean, vvar = vfe_predict(case, umv, [Xuv[i * d:(i + 1) * d] for i in range(len(Xu))])
        tag = "c%d_" % ci
        out[tag + "kind"] = np.array(kind)
        out[tag + "X"], out[tag + "y"], out[tag + "Xs"], out[tag + "u"], out[tag + "Xu"] = pts, yv, Xs, u, Xu
        out[tag + "ls"] = np.array([[0.5] * d, [9.0] * d])
        out[tag + "jitter"] = np.array(1e-5)
        out[tag + "loss"] = np.array(float(loss))
        out[tag + "grad"] = np.array([float(g) for g in grad])
        out[tag + "mean"] = np.array([float(v) for v in mean])
        out[tag + "var"] = np.array([float(v) for v in var])
        out[tag + "vfe_loss"] = np.array(float(vloss))
        out[tag + "vfe_grad"] = np.array([float(g) for g in vgrad])
        out[tag + "vfe_mean"] = np.array([float(v) for v in vmean])
        out[tag + "vfe_var"] = np.array([float(v) for v in vvar])
        print(kind, d, "exact loss %.15g  vfe loss %.15g" % (float(loss), float(vloss)), flush=True)
    out["n_cases"] = np.array(len(specs))
    np.savez(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
