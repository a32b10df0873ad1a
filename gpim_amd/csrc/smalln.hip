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
    long long* prof;       // SMALLN_PROFILE builds only
};

#ifdef SMALLN_PROFILE
#define SSTAMP(i) do { if (threadIdx.x == 0 && it == 5) a.prof[i] = clock64(); } while (0)
#else
#define SSTAMP(i) do { } while (0)
#endif

template <int KIND>
__global__ __launch_bounds__(NTH, 1) void fit_small_kernel(SmallFitArgs a) {
    __shared__ __attribute__((aligned(16))) double D[NB * LDD];
    __shared__ double xr[NB][4];          // raw coordinates (loaded once)
    __shared__ double xs[NB][5];          // scaled coordinates + squared norm
    __shared__ double yv[NB], zv[NB], al[NB], invd[NB];
    __shared__ __attribute__((aligned(16))) double Xs[32 * XS_LD];         // the two scratch tiles of lds_factor_inv
    __shared__ double red[NTH / 64][12];
    __shared__ ThetaDev sth;
    __shared__ double su[MAXP], sm_[MAXP], sv_[MAXP];
    __shared__ double s_prior;
    __shared__ gpimhip_model_t smod;      // LDS copy: lane-indexed bounds would push the argument struct to scratch
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
    if (tid < NB) {
        yv[tid] = (tid < N) ? a.y[tid] : 0.0;
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) xr[tid][k] = (k < d && tid < N) ? a.X[tid * d + k] : 0.0;
    }
    if (tid == 0) {
        s_bad = 0;
        smod = a.m;
    }
    __syncthreads();
    if (tid == 0) {
        ThetaDev t0;
        theta_from_u(smod, su, t0);
        sth = t0;
        s_prior = prior_constant(smod);
    }
    __syncthreads();

    const int niter = a.T > 0 ? a.T : 1;
    for (int it = 0; it < niter; ++it) {
        SSTAMP(0);
        const ThetaDev t = sth;            // theta(u): set before the loop / by the previous finalize
        // Adam scalars of this iteration, fetched now so that the load is long done at finalize
        const double lr_it = a.T > 0 ? a.lr_over_bc1[it] : 0.0, bc2_it = a.T > 0 ? a.bc2_sqrt[it] : 1.0;
        SSTAMP(1);
        if (tid < NB) {
            double s2 = 0.0;
#pragma unroll
            for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
                const double v = (k < d && tid < N) ? xr[tid][k] / t.ls[k] : 0.0;
                xs[tid][k] = v;
                s2 += v * v;
            }
            xs[tid][4] = s2;
        }
        __syncthreads();
        SSTAMP(2);
        // K(X,X) + (jitter + noise) I, lower part of the leading ns x ns block, identity padding.
        // One 16x16 tile per wave and round; with up to three tiles (N <= 32) four waves share a
        // tile, one register group (four rows) each.  Same tile walk in the gradient phase below.
        const int ntile = npan * (npan + 1) / 2;
        const int split = (ntile <= 3) ? 4 : 1;
        for (int w = wave; w < ntile * split; w += NTH / 64) {
            const int q = (split == 4) ? (w >> 2) : w, rg_only = (split == 4) ? (w & 3) : -1;
            int ti = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
            while (ti * (ti + 1) / 2 > q) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= q) ++ti;
            const int tj = q - ti * (ti + 1) / 2;
            const int j = tj * 16 + (lane & 15);
            const double xj0 = xs[j][0], xj1 = xs[j][1], xj2 = xs[j][2], xj3 = xs[j][3], xj4 = xs[j][4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int i = ti * 16 + (lane >> 4) + 4 * rg;
                if (j > i || (rg_only >= 0 && rg != rg_only)) continue;
                double k;
                if (i >= N) {
                    k = (i == j) ? 1.0 : 0.0;
                } else {
                    double dot = xs[i][0] * xj0;
                    dot = fma(xs[i][1], xj1, dot);
                    dot = fma(xs[i][2], xj2, dot);
                    dot = fma(xs[i][3], xj3, dot);
                    const double r2 = clamp0_nan((xs[i][4] - 2.0 * dot) + xj4);
                    k = t.var * kfun_value<KIND>(r2, t.alpha);
                    if (i == j) k += t.diag_add;
                }
                D[i * LDD + j] = k;
            }
        }
        __syncthreads();
        SSTAMP(3);
        // Cholesky and triangular inverse in one sweep (the inverse is built block row by block row
        // behind the factorisation); D holds L^-1 afterwards
        lds_factor_inv(D, invd, Xs, npan, &s_bad, tid, FiNoSink{});
        // not positive-definite: the reference raises at this iteration (gpr.py:192) with u, the Adam
        // state and the history as the previous one left them -- stop here, record how far we got
        __syncthreads();
        if (s_bad != 0) {
            if (tid == 0 && a.T > 0) atomicMin(a.info + 1, it);
            break;
        }
        SSTAMP(4);
        SSTAMP(5);
        SSTAMP(6);
        // z = L^-1 y: four threads per row (columns q, q+4, ...), quad reduction
        {
            const int i = tid >> 2, q = tid & 3;
            double s = 0.0;
            if (i < ns)
                for (int j = q; j <= i; j += 4) s = fma(D[i * LDD + j], yv[j], s);
            s = quad_sum(s);
            if (q == 0 && i < ns) zv[i] = s;
        }
        __syncthreads();
        // alpha = L^-T z: four threads per column
        {
            const int j = tid >> 2, q = tid & 3;
            double s = 0.0;
            if (j < ns)
                for (int i = j + q; i < ns; i += 4) s = fma(D[i * LDD + j], zv[i], s);
            s = quad_sum(s);
            if (q == 0 && j < ns) al[j] = s;
        }
        __syncthreads();
        SSTAMP(7);
        // K^-1 tiles (lower) = sum_{kt >= ti} Linv(kt,ti)^T Linv(kt,tj) on MFMA, consumed in registers
        // by the gradient reduction (same sums as grad_reduce_kernel); components 7, 8 carry
        // |z|^2 and sum log L_ii through the same reduction tree
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (tid < ns) {
            acc[7] = zv[tid] * zv[tid];
            acc[8] = -log(invd[tid]);
        }
        for (int w = wave; w < ntile * split; w += NTH / 64) {
            const int q = (split == 4) ? (w >> 2) : w, rg_only = (split == 4) ? (w & 3) : -1;
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
                if (i >= N || j > i || (rg_only >= 0 && rg != rg_only)) continue;
                const double g = kin[rg] - al[i] * al[j];
                const double wt = (i == j) ? g : 2.0 * g;
                double dot = xs[i][0] * xs[j][0];
                dot = fma(xs[i][1], xs[j][1], dot);
                dot = fma(xs[i][2], xs[j][2], dot);
                dot = fma(xs[i][3], xs[j][3], dot);
                const double r2 = clamp0_nan((xs[i][4] - 2.0 * dot) + xs[j][4]);
                const KVal kv = kfun_grad<KIND>(r2, t.alpha);
                acc[0] = fma(wt, kv.e, acc[0]);
                const double wh = wt * kv.h;
                const double d0 = xs[i][0] - xs[j][0], d1 = xs[i][1] - xs[j][1];
                const double d2_ = xs[i][2] - xs[j][2], d3 = xs[i][3] - xs[j][3];
                acc[1] = fma(wh, d0 * d0, acc[1]);
                acc[2] = fma(wh, d1 * d1, acc[2]);
                acc[3] = fma(wh, d2_ * d2_, acc[3]);
                acc[4] = fma(wh, d3 * d3, acc[4]);
                if (i == j) acc[5] += g;
                if (KIND == GPIMHIP_KERNEL_RQ) acc[6] = fma(wt, kv.ga, acc[6]);
            }
        }
        SSTAMP(8);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const double v = wave_sum(acc[k]);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        SSTAMP(9);
        if (wave == 0) {
            if (lane < 9) {
                double v = 0.0;
                for (int w = 0; w < NTH / 64; ++w) v += red[w][lane];
                red[0][lane] = v;      // only wave 0 touches red here; lanes own distinct slots
            }
            AdamStep st;
            st.beta1 = 0.9; st.beta2 = 0.999; st.eps = 1e-8;
            st.lr_over_bc1 = lr_it;
            st.bc2_sqrt = bc2_it;
            finalize_lanes(smod, N, red[0], lane, &sth, su, sm_, sv_, a.T > 0 ? 1 : 0, st,
                           a.loss ? a.loss + it : nullptr, a.grad,
                           (a.hist && a.T > 0) ? a.hist + (int64_t)it * P : nullptr, s_prior);
        }
        __syncthreads();
        SSTAMP(10);
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
    a.hist = hist; a.loss = loss; a.grad = grad; a.info = h->info; a.prof = nullptr;
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
