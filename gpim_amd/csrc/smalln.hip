// smalln.hip -- fused trainer for N <= 128 observations: the WHOLE Adam loop of
// reconstructor.train (gpim/gpreg/gpr.py:185-199) in ONE launch of ONE workgroup.
//
// This is the Bayesian-optimisation regime (N = 4..60, T = 1000 iterations
// per training, 31 trainings): the general blocked path needs ~10 launches per iteration and is
// purely launch/latency bound there.  Here K, L and L^-1 never leave LDS:
//   per iteration:  u -> theta | K(X,X)+diag -> LDS | panel Cholesky | triangular inverse |
//                   z = L^-1 y, alpha = L^-T z | K^-1 tiles on MFMA fused with the gradient
//                   reduction | loss, chain rule, Adam, history row
// Same arithmetic building blocks as the general path (kfun.hpp, theta.hpp, blocklds.hpp) and the
// same deterministic reduction shapes.
#include "kfun.hpp"
#include "theta.hpp"
#include "blocklds.hpp"

struct SmallFitArgs {
    gpimhip_model_t m;
    const double* X;       // N x d
    int64_t x_bs;          // per-problem stride of X (0: shared)
    const double* y;       // N
    int N, T;
    double* u;             // P, in/out
    const double* lr_over_bc1;   // T  (host-computed: lr / (1 - beta1^t))
    const double* bc2_sqrt;      // T  (sqrt(1 - beta2^t))
    double* hist;          // T x P or null
    double* loss;          // T or null
    double* grad;          // P or null (only when T == 0: evaluate once, no update)
    int32_t* info;
};

template <int KIND>
__global__ __launch_bounds__(NTH, 1) void fit_small_kernel(SmallFitArgs a) {
    __shared__ __attribute__((aligned(16))) double D[NB * LDD];
    __shared__ double xs[NB][5];          // scaled coordinates + squared norm
    __shared__ double yv[NB], zv[NB], al[NB], invd[NB];
    __shared__ double red[NTH / 64][8];
    __shared__ ThetaDev sth;
    __shared__ double su[MAXP], sm_[MAXP], sv_[MAXP];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = a.N, d = a.m.dim;
    const int npan = (N + 15) / 16, ns = npan * 16;
    const int P = 2 + a.m.n_ls + (KIND == GPIMHIP_KERNEL_RQ ? 1 : 0);
    {   // one workgroup per problem of the batch
        const int64_t b = blockIdx.x;
        a.X += b * a.x_bs;
        a.y += b * N;
        a.u += b * P;
        if (a.hist) a.hist += b * a.T * P;
        if (a.loss) a.loss += b * (a.T > 0 ? a.T : 1);
    }

    if (tid < MAXP) {
        su[tid] = (tid < P) ? a.u[tid] : 0.0;
        sm_[tid] = 0.0;
        sv_[tid] = 0.0;
    }
    if (tid < NB) yv[tid] = (tid < N) ? a.y[tid] : 0.0;
    if (tid == 0) s_bad = 0;
    __syncthreads();

    const int niter = a.T > 0 ? a.T : 1;
    for (int it = 0; it < niter; ++it) {
        if (tid == 0) {
            ThetaDev t;
            theta_from_u(a.m, su, t);
            sth = t;
        }
        __syncthreads();
        const ThetaDev t = sth;
        if (tid < NB) {
            double s2 = 0.0;
            for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
                double v = 0.0;
                if (k < d && tid < N) v = a.X[tid * d + k] / t.ls[k];
                xs[tid][k] = v;
                s2 += v * v;
            }
            xs[tid][4] = s2;
        }
        __syncthreads();
        // K(X,X) + (jitter + noise) I, lower part of the leading ns x ns block; identity padding
        for (int e = tid; e < ns * ns; e += NTH) {
            const int i = e / ns, j = e - i * ns;
            if (j > i) continue;
            double k;
            if (i >= N) {
                k = (i == j) ? 1.0 : 0.0;
            } else {
                double dot = xs[i][0] * xs[j][0];
                dot = fma(xs[i][1], xs[j][1], dot);
                dot = fma(xs[i][2], xs[j][2], dot);
                dot = fma(xs[i][3], xs[j][3], dot);
                const double r2 = clamp0_nan((xs[i][4] - 2.0 * dot) + xs[j][4]);
                k = t.var * kfun_value<KIND>(r2, t.alpha);
                if (i == j) k += t.diag_add;
            }
            D[i * LDD + j] = k;
        }
        __syncthreads();
        lds_factor(D, invd, npan, &s_bad, tid);
        // log-determinant (fixed order: one wave, tree)
        double lg = 0.0;
        if (wave == 0) {
            double v = 0.0;
            for (int i = lane; i < ns; i += 64) v -= log(invd[i]);      // log L_ii = -log(1/L_ii)
            v += __shfl_xor(v, 32);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            lg = v;
        }
        // triangular inverse in place: diagonal 16x16 blocks by one wave each, then doubling
        if (wave < npan) trinv16(D + wave * 16 * LDD + wave * 16, LDD, invd + wave * 16,
                                 D + wave * 16 * LDD + wave * 16, LDD, lane);
        __syncthreads();
        lds_invert_levels(D, npan, tid);
        // z = L^-1 y (row-wise), then alpha = L^-T z (column-wise)
        if (tid < ns) {
            double s = 0.0;
            for (int j = 0; j <= tid; ++j) s = fma(D[tid * LDD + j], yv[j], s);
            zv[tid] = s;
        }
        __syncthreads();
        if (tid < ns) {
            double s = 0.0;
            for (int i = tid; i < ns; ++i) s = fma(D[i * LDD + tid], zv[i], s);
            al[tid] = s;
        }
        __syncthreads();
        // K^-1 tiles (lower) = sum_{kt >= ti} Linv(kt,ti)^T Linv(kt,tj) on MFMA, consumed in registers
        // by the gradient reduction (same sums as grad_reduce_kernel)
        double acc7[7] = {0, 0, 0, 0, 0, 0, 0};
        const int ntile = npan * (npan + 1) / 2;
        for (int q = wave; q < ntile; q += NTH / 64) {
            int ti = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
            while (ti * (ti + 1) / 2 > q) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
            const int tj = q - ti * (ti + 1) / 2;
            d4 kin = (d4){0.0, 0.0, 0.0, 0.0};
            for (int kt = ti; kt < npan; ++kt)
#pragma unroll
                for (int s = 0; s < 16; s += 4) {
                    const double av = D[(kt * 16 + s + (lane >> 4)) * LDD + ti * 16 + (lane & 15)];
                    const double bv = D[(kt * 16 + s + (lane >> 4)) * LDD + tj * 16 + (lane & 15)];
                    kin = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, kin, 0, 0, 0);
                }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int i = ti * 16 + (lane >> 4) + 4 * rg, j = tj * 16 + (lane & 15);
                if (i >= N || j > i) continue;
                const double g = kin[rg] - al[i] * al[j];
                const double w = (i == j) ? g : 2.0 * g;
                double dot = xs[i][0] * xs[j][0];
                dot = fma(xs[i][1], xs[j][1], dot);
                dot = fma(xs[i][2], xs[j][2], dot);
                dot = fma(xs[i][3], xs[j][3], dot);
                const double r2 = clamp0_nan((xs[i][4] - 2.0 * dot) + xs[j][4]);
                const KVal kv = kfun_grad<KIND>(r2, t.alpha);
                acc7[0] = fma(w, kv.e, acc7[0]);
                const double wh = w * kv.h;
                const double d0 = xs[i][0] - xs[j][0], d1 = xs[i][1] - xs[j][1];
                const double d2_ = xs[i][2] - xs[j][2], d3 = xs[i][3] - xs[j][3];
                acc7[1] = fma(wh, d0 * d0, acc7[1]);
                acc7[2] = fma(wh, d1 * d1, acc7[2]);
                acc7[3] = fma(wh, d2_ * d2_, acc7[3]);
                acc7[4] = fma(wh, d3 * d3, acc7[4]);
                if (i == j) acc7[5] += g;
                if (KIND == GPIMHIP_KERNEL_RQ) acc7[6] = fma(w, kv.ga, acc7[6]);
            }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            double v = acc7[k];
            v += __shfl_xor(v, 32);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 1);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        if (tid == 0) {
            double Sl[7];
            for (int k = 0; k < 7; ++k) {
                double v = 0.0;
                for (int w = 0; w < NTH / 64; ++w) v += red[w][k];
                Sl[k] = v;
            }
            double q2 = 0.0;
            for (int i = 0; i < ns; ++i) q2 = fma(zv[i], zv[i], q2);
            AdamStep st;
            st.beta1 = 0.9; st.beta2 = 0.999; st.eps = 1e-8;
            st.lr_over_bc1 = a.T > 0 ? a.lr_over_bc1[it] : 0.0;
            st.bc2_sqrt = a.T > 0 ? a.bc2_sqrt[it] : 1.0;
            finalize_step(a.m, N, Sl, q2, lg, t, su, sm_, sv_, a.T > 0 ? 1 : 0, st,
                          a.loss ? a.loss + it : nullptr, a.grad,
                          (a.hist && a.T > 0) ? a.hist + (int64_t)it * P : nullptr);
        }
        __syncthreads();
    }
    if (tid < P && a.T > 0) a.u[tid] = su[tid];
    if (tid == 0 && s_bad != 0 && *a.info == 0) *a.info = s_bad;
}

int launch_fit_small(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, const double* y,
                     int N, double* u, const double* lr_over_bc1, const double* bc2_sqrt, int T, double* hist,
                     double* loss, double* grad) {
    SmallFitArgs a;
    a.m = *m; a.X = X; a.x_bs = x_bs; a.y = y; a.N = N; a.T = T; a.u = u;
    a.lr_over_bc1 = lr_over_bc1; a.bc2_sqrt = bc2_sqrt;
    a.hist = hist; a.loss = loss; a.grad = grad; a.info = h->info;
    switch (m->kernel) {
        case GPIMHIP_KERNEL_RBF:
            hipLaunchKernelGGL((fit_small_kernel<GPIMHIP_KERNEL_RBF>), dim3(h->nbatch), dim3(NTH), 0, h->stream, a); break;
        case GPIMHIP_KERNEL_MATERN52:
            hipLaunchKernelGGL((fit_small_kernel<GPIMHIP_KERNEL_MATERN52>), dim3(h->nbatch), dim3(NTH), 0, h->stream, a); break;
        case GPIMHIP_KERNEL_RQ:
            hipLaunchKernelGGL((fit_small_kernel<GPIMHIP_KERNEL_RQ>), dim3(h->nbatch), dim3(NTH), 0, h->stream, a); break;
        default: gpim_set_error("unknown kernel kind"); return GPIMHIP_E_BADARG;
    }
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
