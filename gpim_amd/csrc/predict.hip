// predict.hip -- fused posterior (+ acquisition) kernel for models with few observations
// (the "K* never written to HBM" design; replaces gpmodel.predict inside
// the acquisition functions, gpim/gpbayes/acqfunc.py:27-29,58-60,86-88, and GPRegression.forward,
// gpim/gpreg/gpr.py:243-250, when N <= 384).
//
// One workgroup owns CT = 32 test points.  It
//   1. builds its K(X, X*) panel (np x 32) in LDS with the arithmetic of kmat_kernel (same bits as the slab
//      the large-N path writes to HBM),
//   2. mean_c = sum_k K*[k][c] alpha[k]                         (fixed-shape 8-way split, LDS reduce),
//   3. W = L^-1 K* on v_mfma_f64_16x16x4_f64: every wave takes 16-row tiles of L^-1 (fragments straight
//      from L2 -- L^-1 is at most 1.2 MB and shared by all workgroups), both 16-column tiles of the panel;
//      only the column sums of squares of W are kept,
//   4. var_c = clamp(s2 - sum_i W[i][c]^2, 0) + noise, and optionally sd, CB / EI / POI (incumbent read
//      from device memory, so the observed-rows prediction -> nanmax -> grid sweep chain needs no host
//      round trip) and the NaN mask.
// The large-N path keeps the slab: there every 128-row tile of L^-1 K* would have to regenerate the K*
// operand of its whole k-range (N/256 times the exp() work per element on average) inside a loop that is
// MFMA-bound, while writing and re-reading the slab costs 1.5 % of the product's time.
#include "kfun.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));

#define PF_CT 32
#define PF_LDK (PF_CT + 16)     // row stride of the K* panel in LDS: 4 k-rows of a B fragment hit distinct banks

struct FusedPredictArgs {
    const double* X; int64_t N; int64_t x_bs; int d;
    const double* Xs; int64_t M;
    const ThetaDev* th;
    const double* Linv; int64_t ld; int64_t np;
    const double* alpha;
    double* mean_out; double* var_out;           // M each (per problem), var_out may be null when sd_out is given
    double* sd_out; double* acq_out;             // optional (single problem only)
    const double* mask; const double* p0_dev;
    int acq_kind; double p0, p1;
};

template <int KIND>
__global__ __launch_bounds__(256, 2) void predict_fused_kernel(FusedPredictArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int np = (int)a.np;
    double* Ks = smem;                                  // [np][PF_LDK]
    double* xa = smem + (size_t)np * PF_LDK;            // [np][5], later red[4][32] + part[8][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t c0 = (int64_t)blockIdx.x * PF_CT;
    const int b = blockIdx.y;
    const double* X = a.X + b * a.x_bs;
    const ThetaDev t = a.th[b];
    const double* Linv = a.Linv + (int64_t)b * a.np * a.ld;
    const double* alpha = a.alpha + (int64_t)b * a.np;

    // scaled training coordinates (as kmat_kernel: division by the lengthscale, |a|^2 in slot 4)
    for (int g = tid; g < np; g += 256) {
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
            double v = 0.0;
            if (k < a.d && g < a.N) v = X[(int64_t)g * a.d + k] / t.ls[k];
            xa[g * 5 + k] = v;
            s2 += v * v;
        }
        xa[g * 5 + 4] = s2;
    }
    // this thread's test point (column c) in scaled coordinates
    const int c = tid & (PF_CT - 1), part = tid >> 5;       // 8 row groups
    const int64_t gj = c0 + c;
    double z[GPIMHIP_MAX_DIM], zn = 0.0;
#pragma unroll
    for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
        double v = 0.0;
        if (k < a.d && gj < a.M) v = a.Xs[gj * a.d + k] / t.ls[k];
        z[k] = v;
        zn += v * v;
    }
    __syncthreads();
    double msum = 0.0;
    for (int g = part; g < np; g += 8) {
        double dot = xa[g * 5 + 0] * z[0];
        dot = fma(xa[g * 5 + 1], z[1], dot);
        dot = fma(xa[g * 5 + 2], z[2], dot);
        dot = fma(xa[g * 5 + 3], z[3], dot);
        double r2 = (xa[g * 5 + 4] - 2.0 * dot) + zn;
        r2 = clamp0_nan(r2);
        double k = t.var * kfun_value<KIND>(r2, t.alpha);
        if (g >= a.N || gj >= a.M) k = 0.0;
        Ks[g * PF_LDK + c] = k;
        msum = fma(k, alpha[g], msum);
    }
    __syncthreads();                                    // K* panel complete; xa is free
    double* red = xa;                                   // [4][32]
    double* mpart = xa + 4 * PF_CT;                     // [8][32]
    mpart[part * PF_CT + c] = msum;

    // W = L^-1 K*: wave w takes row tiles w, w+4, ...; accumulators for the two 16-column tiles
    double ssq0 = 0.0, ssq1 = 0.0;
    const int nrt = np >> 4;
    const int ar = lane & 15, ak = lane >> 4;
    for (int rt = wave; rt < nrt; rt += 4) {
        d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
        const double* arow = Linv + (int64_t)(rt * 16 + ar) * a.ld + ak;
        double av[4], an[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) av[kk] = arow[kk * 4];
        for (int kb = 0; kb <= rt; ++kb) {
            if (kb < rt) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) an[kk] = arow[(kb + 1) * 16 + kk * 4];
            }
            const double* bp = Ks + (kb * 16 + ak) * PF_LDK + ar;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double b0 = bp[kk * 4 * PF_LDK], b1 = bp[kk * 4 * PF_LDK + 16];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], b1, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) av[kk] = an[kk];
        }
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            ssq0 = fma(acc0[rg], acc0[rg], ssq0);
            ssq1 = fma(acc1[rg], acc1[rg], ssq1);
        }
    }
    // D layout: column = lane & 15 -> add the four lane groups, then the four waves (fixed order)
    ssq0 += __shfl_xor(ssq0, 16);
    ssq0 += __shfl_xor(ssq0, 32);
    ssq1 += __shfl_xor(ssq1, 16);
    ssq1 += __shfl_xor(ssq1, 32);
    if (lane < 16) {
        red[wave * PF_CT + lane] = ssq0;
        red[wave * PF_CT + 16 + lane] = ssq1;
    }
    __syncthreads();
    if (tid < PF_CT && c0 + tid < a.M) {
        const int64_t j = c0 + tid;
        double q = red[tid];
        q += red[PF_CT + tid];
        q += red[2 * PF_CT + tid];
        q += red[3 * PF_CT + tid];
        double mu = mpart[tid];
#pragma unroll
        for (int p = 1; p < 8; ++p) mu += mpart[p * PF_CT + tid];
        const double var = clamp0_nan(t.var - q) + t.noise;
        a.mean_out[(int64_t)b * a.M + j] = mu;
        if (a.var_out) a.var_out[(int64_t)b * a.M + j] = var;
        if (a.sd_out || a.acq_out) {
            const double s = sqrt(var);
            if (a.sd_out) a.sd_out[j] = s;
            if (a.acq_out) {
                const double p0 = a.p0_dev ? *a.p0_dev : a.p0;
                double v;
                if (a.acq_kind == GPIMHIP_ACQ_CB) {
                    v = p0 * mu + a.p1 * s;
                } else {
                    const double imp = mu - p0 - a.p1;
                    const double zz = imp / s;
                    const double cdf = 0.5 * erfc(-zz * 0.7071067811865476);
                    if (a.acq_kind == GPIMHIP_ACQ_EI) {
                        const double pdf = exp(-0.5 * zz * zz) * 0.3989422804014327;
                        v = imp * cdf + s * pdf;
                    } else {
                        v = cdf;
                    }
                }
                if (a.mask) v = a.mask[j] * v;
                a.acq_out[j] = v;
            }
        }
    }
}

// sd = sqrt(var) and the acquisition sweep for the slab path (N > 384): same formulas as acq_kernel
// (engine.hip), incumbent from device memory
__global__ void acq_from_var_kernel(int kind, const double* __restrict__ mean, const double* __restrict__ var, int64_t M,
                                    double p0, const double* __restrict__ p0_dev, double p1,
                                    const double* __restrict__ mask, double* __restrict__ sd_out,
                                    double* __restrict__ acq_out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double mu = mean[j], s = sqrt(var[j]);
    if (sd_out) sd_out[j] = s;
    if (!acq_out) return;
    if (p0_dev) p0 = *p0_dev;
    double v;
    if (kind == GPIMHIP_ACQ_CB) {
        v = p0 * mu + p1 * s;
    } else {
        const double imp = mu - p0 - p1;
        const double zz = imp / s;
        const double cdf = 0.5 * erfc(-zz * 0.7071067811865476);
        if (kind == GPIMHIP_ACQ_EI) {
            const double pdf = exp(-0.5 * zz * zz) * 0.3989422804014327;
            v = imp * cdf + s * pdf;
        } else {
            v = cdf;
        }
    }
    if (mask) v = mask[j] * v;
    acq_out[j] = v;
}
int launch_acq_from_var(gpimhip_ctx* h, int kind, const double* mean, const double* var, int64_t M, double p0,
                        const double* p0_dev, double p1, const double* mask, double* sd_out, double* acq_out) {
    hipLaunchKernelGGL(acq_from_var_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, kind, mean, var,
                       M, p0, p0_dev, p1, mask, sd_out, acq_out);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

bool fused_predict_fits(int64_t np) {
    return np <= 384 && !getenv("GPIMHIP_NO_FUSED_PREDICT");
}

// mean / var (and optionally sd, acquisition) at M test points from the factorised model held in the
// workspace (theta, L^-1 in A, alpha).  acq_kind < 0: no acquisition.
int launch_predict_fused(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, int64_t N,
                         const double* Xs, int64_t M, double* mean_out, double* var_out, double* sd_out,
                         double* acq_out, int acq_kind, double p0, const double* p0_dev, double p1,
                         const double* mask) {
    FusedPredictArgs a;
    a.X = X; a.N = N; a.x_bs = x_bs; a.d = m->dim;
    a.Xs = Xs; a.M = M; a.th = h->theta;
    a.Linv = h->A; a.ld = h->ld; a.np = h->np; a.alpha = h->alpha;
    a.mean_out = mean_out; a.var_out = var_out; a.sd_out = sd_out; a.acq_out = acq_out;
    a.mask = mask; a.p0_dev = p0_dev; a.acq_kind = acq_kind; a.p0 = p0; a.p1 = p1;
    const size_t lds = ((size_t)h->np * PF_LDK + std::max<size_t>((size_t)h->np * 5, 12 * PF_CT)) * sizeof(double);
    const dim3 grid((unsigned)((M + PF_CT - 1) / PF_CT), h->nbatch);
#define PF_LAUNCH(KIND)                                                                                            \
    do {                                                                                                           \
        /* once per process and kernel; several host threads may predict at the same time (dist.reconstruct_slices) */ \
        static const hipError_t attr_rc = hipFuncSetAttribute((const void*)predict_fused_kernel<KIND>,             \
                                                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        HIP_TRY(attr_rc);                                                                                          \
        hipLaunchKernelGGL(predict_fused_kernel<KIND>, grid, dim3(256), lds, h->stream, a);                        \
    } while (0)
    if (m->kernel == GPIMHIP_KERNEL_RBF) PF_LAUNCH(GPIMHIP_KERNEL_RBF);
    else if (m->kernel == GPIMHIP_KERNEL_MATERN52) PF_LAUNCH(GPIMHIP_KERNEL_MATERN52);
    else PF_LAUNCH(GPIMHIP_KERNEL_RQ);
#undef PF_LAUNCH
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
