// kfun.hpp -- the covariance functions and their parameter derivatives (device side).
//
// Restates pyro.contrib.gp.kernels.{RBF,Matern52,RationalQuadratic} as used through
// gpim/kernels/pyro_kernels.py:59-67 (formulas: SURVEY App. A.3) with the SAME algebraic form
// as the reference so that round-off behaves alike:
//   a = x / l, b = z / l;  r2 = max((|a|^2 - 2 a.b) + |b|^2, 0)   (NaN propagates like torch.clamp)
//   RBF       k = s2 * exp(-r2/2)
//   Matern52  r = sqrt(r2 + 1e-12);  k = s2 * (1 + sqrt5*r + (5/3) r2) * exp(-sqrt5*r)
//             The reading followed (pyro-ppl 1.x, pyro/contrib/gp/kernels/isotropic.py, unchanged since 0.3; the package is
//             not in this image and the reference declares it unpinned, setup.py:30): Matern52.forward computes
//               r2 = self._square_scaled_dist(X, Z);  r = _torch_sqrt(r2)   # (r2 + 1e-12).sqrt()
//               sqrt5_r = 5**0.5 * r;  return variance * (1 + sqrt5_r + (5/3) * r2) * exp(-sqrt5_r)
//             i.e. the shift enters through r only and the (5/3) term takes the UN-shifted squared distance (rounds 1-5
//             used r*r there: <= 1.7e-12 s2 per entry, below every bar; oracle, mpmath fixtures and this file changed
//             together in round 6).  dk/dr2 = -(5/6) (1 + sqrt5 r2 / r) exp(-sqrt5 r).
//   RQ        k = s2 * (1 + r2/(2*alpha))^(-alpha)
#pragma once
#include "common.hpp"

#define SQRT5 2.23606797749979

struct KVal {
    double e;      // k / s2
    double h;      // (dk/dl_k) = s2 * h * (a_k - b_k)^2 / l_k       (per-dimension factor)
    double ga;     // (dk/dalpha) / s2                                 (RQ only)
};

// exp(x) for x <= 0 and sqrt(x) for 1e-13 <= x < 1e300, the only arguments the covariance functions have (round 6).  The
// device library's fp64 exp / sqrt expand to ~40 / ~22 instructions each -- range checks for arguments that cannot occur
// here, denormal scaling, and a register copy in front of every step of the polynomial -- and the kernel-matrix build and
// the gradient contraction are bound by exactly these instructions (~85 per Matern52 entry: 3.6 / 2.8 TB/s of the 8 an
// HBM-bound pass would reach).  Same algorithms, nothing else:
//   kf_exp_neg   n = rint(x log2 e), r = x - n ln2 (two-part), degree-11 polynomial on |r| <= ln2 / 2 (interpolated at the
//                Chebyshev nodes: 4.2e-18 relative in exact arithmetic), ldexp(p, n) (underflows to 0 by itself; NaN stays NaN)
//   kf_sqrt      v_rsq_f64 seed, one coupled Newton step on (g, h) = (sqrt x, 1 / (2 sqrt x)), one correction of g
// each within 1 ulp of the library's result (tools/r6_kfun_probe.hip: max 1 ulp, mean 0.07 over 2e7 arguments).
// The polynomial's steps are written as three-address v_fma_f64 with the coefficient in a scalar register: the compiler's
// own choice (v_fmac into a fresh copy of the coefficient) doubles the instruction count of the chain.
__device__ __forceinline__ double kf_fma_c(double a, double b, double c) {      // a * b + c, c wave-uniform
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ double kf_exp_neg(double x) {
    x = (x < -746.0) ? -746.0 : x;      // exp(-746) = 0 in double; a lengthscale near its lower bound 0 gives -inf here (NaN stays NaN)
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = fma(n, -0.6931471803691238, x);
    r = fma(n, -1.9082149292705877e-10, r);
    double p = kf_fma_c(r, 2.5110037605963777e-08, 2.763263963904103e-07);
    p = kf_fma_c(p, r, 2.755724091857897e-06);
    p = kf_fma_c(p, r, 2.4801485482328494e-05);
    p = kf_fma_c(p, r, 0.00019841269890047113);
    p = kf_fma_c(p, r, 0.0013888888952314775);
    p = kf_fma_c(p, r, 0.008333333333319601);
    p = kf_fma_c(p, r, 0.0416666666664881);
    p = kf_fma_c(p, r, 0.1666666666666668);
    p = kf_fma_c(p, r, 0.5000000000000019);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
__device__ __forceinline__ double kf_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}

template <int KIND>
__device__ __forceinline__ double kfun_value(double r2, double alpha) {
    if (KIND == GPIMHIP_KERNEL_RBF) return kf_exp_neg(-0.5 * r2);
    if (KIND == GPIMHIP_KERNEL_MATERN52) {
        const double r = kf_sqrt(r2 + 1e-12);
        const double s5r = SQRT5 * r;
        return (1.0 + s5r + (5.0 / 3.0) * r2) * kf_exp_neg(-s5r);
    }
    return pow(1.0 + (0.5 / alpha) * r2, -alpha);
}

template <int KIND>
__device__ __forceinline__ KVal kfun_grad(double r2, double alpha) {
    KVal v;
    v.ga = 0.0;
    if (KIND == GPIMHIP_KERNEL_RBF) {
        v.e = kf_exp_neg(-0.5 * r2);
        v.h = v.e;
    } else if (KIND == GPIMHIP_KERNEL_MATERN52) {
        const double r = kf_sqrt(r2 + 1e-12);
        const double s5r = SQRT5 * r;
        const double ex = kf_exp_neg(-s5r);
        v.e = (1.0 + s5r + (5.0 / 3.0) * r2) * ex;
        v.h = (5.0 / 3.0) * (1.0 + SQRT5 * (r2 / r)) * ex;
    } else {
        const double t = (0.5 / alpha) * r2;
        const double base = 1.0 + t;
        v.e = pow(base, -alpha);
        v.h = v.e / base;
        v.ga = v.e * (t / base - log1p(t));
    }
    return v;
}

__device__ __forceinline__ double clamp0_nan(double v) { return (v < 0.0) ? 0.0 : v; }
