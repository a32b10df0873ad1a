// theta.hpp -- hyper-parameter maps and the scalar tail of one training iteration (device side),
// shared by the general path (engine.hip) and the fused small-N trainer (smalln.hip).
//
//   u -> theta : torch.distributions transform_to(interval) = Affine o Sigmoid with clipping,
//                transform_to(positive) = exp  (pyro_kernels.py:81-94)
//   finalize   : loss, chain rule d loss/du, one torch.optim.Adam step, history row
//                (gpr.py:185-199)
#pragma once
#include "common.hpp"

__device__ __forceinline__ void interval_map(double u, double lo, double hi, double& val, double& dval) {
    const double tiny = 2.2250738585072014e-308, eps = 2.220446049250313e-16;
    double s = 1.0 / (1.0 + exp(-u));
    double ds = s * (1.0 - s);
    if (s < tiny) { s = tiny; ds = 0.0; }
    if (s > 1.0 - eps) { s = 1.0 - eps; ds = 0.0; }
    val = lo + (hi - lo) * s;
    dval = (hi - lo) * ds;
}

__device__ inline void theta_from_u(const gpimhip_model_t& m, const double* u, ThetaDev& t) {
    interval_map(u[0], m.amp_lo, m.amp_hi, t.var, t.dvar_du);
    for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
        const int src = (m.n_ls == 1) ? 0 : k;
        if (k < m.dim) {
            interval_map(u[1 + src], m.ls_lo[src], m.ls_hi[src], t.ls[k], t.dls_du[k]);
        } else {
            t.ls[k] = 1.0;
            t.dls_du[k] = 0.0;
        }
        t.inv_ls[k] = 1.0 / t.ls[k];
    }
    t.noise = exp(u[1 + m.n_ls]);
    t.dnoise_du = t.noise;
    if (m.kernel == GPIMHIP_KERNEL_RQ) {
        t.alpha = exp(u[2 + m.n_ls]);
        t.dalpha_du = t.alpha;
    } else {
        t.alpha = 1.0;
        t.dalpha_du = 0.0;
    }
    t.diag_add = m.jitter + t.noise;
}


// S[0..6]: reduced gradient sums (see grad_reduce_kernel), q2 = |L^-1 y|^2, lg = sum log L_ii
__device__ inline double prior_constant(const gpimhip_model_t& m) {
    // -log prior of the Uniform priors (constant inside their support)
    double prior = log(m.amp_hi - m.amp_lo);
    for (int k = 0; k < m.n_ls; ++k) prior += log(m.ls_hi[k] - m.ls_lo[k]);
    return prior;
}

// g (MAXP doubles) and tn: working storage -- private arrays in finalize_step, LDS where the step is the tail of a kernel
// whose other waves must not pay for scratch memory (grad_reduce_kernel).
// theta_next (optional): theta of the STEPPED parameters -- what the next iteration's theta launch would compute from u
// (the same theta_from_u); `t` must not alias it.
__device__ __forceinline__ void finalize_step_ws(const gpimhip_model_t& m, int64_t N, const double* S, double q2, double lg,
                                                 const ThetaDev& t, double* u, double* adam_m, double* adam_v, int do_adam,
                                                 const AdamStep& st, double* loss_out, double* grad_out, double* hist_row,
                                                 double prior, ThetaDev* theta_next, double* g, ThetaDev& tn) {
    const int P = 2 + m.n_ls + (m.kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    const double loss = 0.5 * q2 + lg + 0.5 * (double)N * 1.8378770664093453 + prior;
    g[0] = 0.5 * S[0] * t.dvar_du;
    if (m.n_ls == 1) {
        double s = 0.0;
        for (int k = 0; k < m.dim; ++k) s += S[1 + k];
        g[1] = 0.5 * s * t.var / t.ls[0] * t.dls_du[0];
    } else {
        for (int k = 0; k < m.dim; ++k) g[1 + k] = 0.5 * S[1 + k] * t.var / t.ls[k] * t.dls_du[k];
    }
    g[1 + m.n_ls] = 0.5 * S[5] * t.dnoise_du;
    if (m.kernel == GPIMHIP_KERNEL_RQ) g[2 + m.n_ls] = 0.5 * S[6] * t.var * t.dalpha_du;
    if (loss_out) *loss_out = loss;
    if (grad_out)
        for (int k = 0; k < P; ++k) grad_out[k] = g[k];
    if (do_adam) {
        for (int k = 0; k < P; ++k) {
            double mm = adam_m[k], vv = adam_v[k];
            mm = mm + (g[k] - mm) * (1.0 - st.beta1);            // exp_avg.lerp_(grad, 1 - beta1)
            vv = vv * st.beta2 + (1.0 - st.beta2) * g[k] * g[k]; // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
            const double denom = sqrt(vv) / st.bc2_sqrt + st.eps;
            u[k] = u[k] + (-st.lr_over_bc1) * (mm / denom);      // param.addcdiv_(exp_avg, denom, value=-step_size)
            adam_m[k] = mm;
            adam_v[k] = vv;
        }
        if (hist_row || theta_next) {
            theta_from_u(m, u, tn);
            if (hist_row) {
                hist_row[0] = tn.var;
                for (int k = 0; k < m.n_ls; ++k) hist_row[1 + k] = tn.ls[k];
                hist_row[1 + m.n_ls] = tn.noise;
                if (m.kernel == GPIMHIP_KERNEL_RQ) hist_row[2 + m.n_ls] = tn.alpha;
            }
            if (theta_next) *theta_next = tn;
        }
    }
}
__device__ inline void finalize_step(const gpimhip_model_t& m, int64_t N, const double* S, double q2, double lg,
                                     const ThetaDev& t, double* u, double* adam_m, double* adam_v, int do_adam,
                                     const AdamStep& st, double* loss_out, double* grad_out, double* hist_row,
                                     double prior, ThetaDev* theta_next = nullptr) {
    double g[MAXP];
    ThetaDev tn;
    finalize_step_ws(m, N, S, q2, lg, t, u, adam_m, adam_v, do_adam, st, loss_out, grad_out, hist_row, prior, theta_next, g, tn);
}

// Lane-parallel form of finalize_step for the fused small-N trainer: lane p of one wave owns
// hyper-parameter p (chain rule, Adam step, theta of the stepped value), so the exp/divide/sqrt
// chains of the P parameters run side by side.  Same expressions, in the same order, as above.
// `t` is both the current theta (read) and the next one (written) -- it lives in LDS.
__device__ inline void finalize_lanes(const gpimhip_model_t& m, int64_t N, const double* S, int lane, ThetaDev* t,
                                      double* u, double* adam_m, double* adam_v, int do_adam, const AdamStep& st,
                                      double* loss_out, double* grad_out, double* hist_row, double prior) {
    const int P = 2 + m.n_ls + (m.kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    if (lane == 0 && loss_out) *loss_out = 0.5 * S[7] + S[8] + 0.5 * (double)N * 1.8378770664093453 + prior;
    if (lane >= P) return;
    const bool is_var = lane == 0, is_ls = lane >= 1 && lane <= m.n_ls, is_noise = lane == 1 + m.n_ls;
    const int k = is_ls ? lane - 1 : 0;
    double g;
    if (is_var) {
        g = 0.5 * S[0] * t->dvar_du;
    } else if (is_ls) {
        double s = S[1 + k];
        if (m.n_ls == 1) {
            s = 0.0;
            for (int q = 0; q < m.dim; ++q) s += S[1 + q];
        }
        g = 0.5 * s * t->var / t->ls[k] * t->dls_du[k];
    } else if (is_noise) {
        g = 0.5 * S[5] * t->dnoise_du;
    } else {
        g = 0.5 * S[6] * t->var * t->dalpha_du;
    }
    if (grad_out) grad_out[lane] = g;
    if (!do_adam) return;
    double mm = adam_m[lane], vv = adam_v[lane];
    mm = mm + (g - mm) * (1.0 - st.beta1);
    vv = vv * st.beta2 + (1.0 - st.beta2) * g * g;
    const double denom = sqrt(vv) / st.bc2_sqrt + st.eps;
    const double un = u[lane] + (-st.lr_over_bc1) * (mm / denom);
    u[lane] = un;
    adam_m[lane] = mm;
    adam_v[lane] = vv;
    double val, dval;
    if (is_var || is_ls) {
        interval_map(un, is_var ? m.amp_lo : m.ls_lo[k], is_var ? m.amp_hi : m.ls_hi[k], val, dval);
    } else {
        val = exp(un);
        dval = val;
    }
    if (hist_row) hist_row[lane] = val;
    if (is_var) {
        t->var = val;
        t->dvar_du = dval;
    } else if (is_ls) {
        const double inv = 1.0 / val;
        if (m.n_ls == 1) {
            for (int q = 0; q < m.dim; ++q) { t->ls[q] = val; t->dls_du[q] = dval; t->inv_ls[q] = inv; }
        } else {
            t->ls[k] = val; t->dls_du[k] = dval; t->inv_ls[k] = inv;
        }
    } else if (is_noise) {
        t->noise = val;
        t->dnoise_du = val;
        t->diag_add = m.jitter + val;
    } else {
        t->alpha = val;
        t->dalpha_du = val;
    }
}
