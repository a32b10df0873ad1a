// engine.hip -- the non-GEMM device kernels of the exact-GP hot path:
//   theta_kernel        u -> constrained hyper-parameters + chain-rule factors       (row a4)
//   kmat_kernel         LDS-tiled pairwise-distance + covariance build               (row a5)
//   trmv / trmv_t       z = L^-1 y, alpha = L^-T z                                   (row a7)
//   grad_reduce_kernel  fused  sum_ij (K^-1 - alpha alpha^T)_ij dK_ij/dtheta          (row a8)
//   finalize_kernel     loss, d loss/du, Adam step, history row                      (rows a7-a10)
//   predict helpers     mean = K*^T alpha, var = clamp(s2 - colsumsq, 0) + noise     (row a11)
//   acq / nanmax / topk acquisition sweep and ranking                                (rows a13, a14)
// Row labels refer to SURVEY.md section 8(a).  All reductions use fixed-shape trees so results are
// bit-reproducible run to run.
#include "kfun.hpp"
#include <cstring>
#include <type_traits>
#include "theta.hpp"

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
// The N x N matrices (covariance, factor, inverse, K* slab) are double or float (gpimhip_set_precision); vectors,
// reductions and every scalar stay double.  Host-side the pointers are typed double* for both.
template <typename R> struct Vec2;
template <> struct Vec2<double> { typedef d2 T; };
template <> struct Vec2<float> { typedef f2 T; };

// ------------------------------------------------------------------------------------------
// u -> theta   (torch.distributions transform_to(interval) = Affine o Sigmoid with clipping;
//               transform_to(positive) = exp.  SURVEY App. A.2)
// ------------------------------------------------------------------------------------------
// Batch convention of this file: blockIdx.y = problem index b; every per-problem pointer advances by
// b times the stride passed next to it (workspace buffers of the B problems are stacked).
__global__ void theta_kernel(gpimhip_model_t m, const double* __restrict__ u, int ustride,
                             ThetaDev* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ThetaDev t;
        theta_from_u(m, u + (int64_t)blockIdx.y * ustride, t);
        out[blockIdx.y] = t;
    }
}

// raw = [s2, l_0.., alpha]  (operator-level gpimhip_kmat)
__global__ void theta_raw_kernel(gpimhip_model_t m, const double* __restrict__ raw, ThetaDev* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ThetaDev t;
        t.var = raw[0];
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
            t.ls[k] = (k < m.dim) ? raw[1 + ((m.n_ls == 1) ? 0 : k)] : 1.0;
            t.inv_ls[k] = 1.0 / t.ls[k];
            t.dls_du[k] = 0.0;
        }
        t.alpha = (m.kernel == GPIMHIP_KERNEL_RQ) ? raw[1 + m.n_ls] : 1.0;
        t.noise = 0.0;
        t.diag_add = 0.0;
        t.dvar_du = t.dnoise_du = t.dalpha_du = 0.0;
        *out = t;
    }
}

int launch_theta(gpimhip_ctx* h, const gpimhip_model_t* m, const double* u) {
    const int P = 2 + m->n_ls + (m->kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    hipLaunchKernelGGL(theta_kernel, dim3(1, h->nbatch), dim3(64), 0, h->stream, *m, u, P, h->theta);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
int launch_theta_raw(gpimhip_ctx* h, const gpimhip_model_t* m, const double* raw) {
    hipLaunchKernelGGL(theta_raw_kernel, dim3(1), dim3(64), 0, h->stream, *m, raw, h->theta1);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// covariance build.  One workgroup = one 128x128 tile; the 128 + 128 scaled coordinate rows are
// staged once in LDS, every thread produces an 8x8 patch and stores 16-byte pairs so that each
// wave writes 4 rows x 256 contiguous bytes.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lower_tile_from_linear(int q, int& i, int& j) {
    i = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while ((int64_t)i * (i + 1) / 2 > q) --i;
    while ((int64_t)(i + 1) * (i + 2) / 2 <= q) ++i;
    j = q - (int)((int64_t)i * (i + 1) / 2);
}

template <int KIND, typename R>
__global__ __launch_bounds__(256) void kmat_kernel(const double* __restrict__ X, int64_t N,
                                                   const double* __restrict__ Z, int64_t M, int d,
                                                   const ThetaDev* __restrict__ th, double diag_add,
                                                   int use_theta_diag, R* __restrict__ out, int64_t ld,
                                                   int ntc, int sym, int lower_only, int64_t x_bs, int64_t z_bs,
                                                   int64_t out_bs) {
    __shared__ double xa[128][5];
    __shared__ double xz[128][5];
    const int tid = threadIdx.x;
    X += blockIdx.y * x_bs;
    Z += blockIdx.y * z_bs;
    th += blockIdx.y;
    out += blockIdx.y * out_bs;
    int ci, cj;
    if (lower_only) lower_tile_from_linear(blockIdx.x, ci, cj);
    else { ci = blockIdx.x / ntc; cj = blockIdx.x % ntc; }
    const ThetaDev t = *th;
    {
        const bool isrow = tid < 128;
        const int loc = tid & 127;
        const int64_t g = (int64_t)(isrow ? ci : cj) * 128 + loc;
        const double* src = isrow ? X : Z;
        const int64_t lim = isrow ? N : M;
        double s2 = 0.0;
        double (*dst)[5] = isrow ? xa : xz;
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
            double a = 0.0;
            if (k < d && g < lim) a = src[g * d + k] / t.ls[k];
            dst[loc][k] = a;
            s2 += a * a;
        }
        dst[loc][4] = s2;
    }
    __syncthreads();
    const double dadd = use_theta_diag ? t.diag_add : diag_add;
    const int ty = tid >> 4, tx = tid & 15;
    // a tile that lies inside the matrix and off its diagonal (all but 2 nb - 1 of the nb (nb + 1) / 2 lower tiles) needs
    // no per-entry bounds / diagonal tests (64-bit compares: a tenth of the entry's instructions)
    const bool interior = (int64_t)ci * 128 + 128 <= N && (int64_t)cj * 128 + 128 <= M && !(sym && ci == cj);
    auto body = [&](auto INTERIOR) {
#pragma unroll 2
        for (int rr = 0; rr < 8; ++rr) {
            const int r = ty + 16 * rr;
            const int64_t gi = (int64_t)ci * 128 + r;
            const double a0 = xa[r][0], a1 = xa[r][1], a2 = xa[r][2], a3 = xa[r][3], an = xa[r][4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                typename Vec2<R>::T v;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int c = tx * 2 + 32 * cc + e;
                    const int64_t gj = (int64_t)cj * 128 + c;
                    double dot = a0 * xz[c][0];
                    dot = fma(a1, xz[c][1], dot);
                    dot = fma(a2, xz[c][2], dot);
                    dot = fma(a3, xz[c][3], dot);
                    double r2 = (an - 2.0 * dot) + xz[c][4];
                    r2 = clamp0_nan(r2);
                    double k = t.var * kfun_value<KIND>(r2, t.alpha);
                    if (!decltype(INTERIOR)::value) {
                        if (gi >= N || gj >= M) k = (sym && gi == gj) ? 1.0 : 0.0;
                        else if (sym && gi == gj) k += dadd;
                    }
                    v[e] = (R)k;
                }
                *reinterpret_cast<typename Vec2<R>::T*>(out + gi * ld + (int64_t)cj * 128 + tx * 2 + 32 * cc) = v;
            }
        }
    };
    if (interior) body(std::true_type{});
    else body(std::false_type{});
}

int launch_kmat(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t N, const double* Z,
                int64_t M, const ThetaDev* theta, double diag_add, int use_theta_diag, double* out,
                int64_t ld, int64_t rows_pad, int64_t cols_pad, int sym, int lower_only, int64_t x_bs,
                int64_t z_bs, int64_t out_bs) {
    const int ntr = (int)(rows_pad / 128), ntc = (int)(cols_pad / 128);
    const int64_t nblk = lower_only ? (int64_t)ntr * (ntr + 1) / 2 : (int64_t)ntr * ntc;
    if (nblk <= 0) return GPIMHIP_OK;
    const double* Zp = Z ? Z : X;
    dim3 grid((unsigned)nblk, h->nbatch), block(256);
    if (!Z) z_bs = x_bs;
#define KM_LAUNCH(KIND)                                                                              \
    do {                                                                                             \
        if (h->fp32)                                                                                 \
            hipLaunchKernelGGL((kmat_kernel<KIND, float>), grid, block, 0, h->stream, X, N, Zp, M, m->dim, theta, diag_add, \
                               use_theta_diag, reinterpret_cast<float*>(out), ld, ntc, sym, lower_only, x_bs, z_bs, out_bs); \
        else                                                                                         \
            hipLaunchKernelGGL((kmat_kernel<KIND, double>), grid, block, 0, h->stream, X, N, Zp, M, m->dim, theta, diag_add, \
                               use_theta_diag, out, ld, ntc, sym, lower_only, x_bs, z_bs, out_bs);     \
    } while (0)
    switch (m->kernel) {
        case GPIMHIP_KERNEL_RBF: KM_LAUNCH(GPIMHIP_KERNEL_RBF); break;
        case GPIMHIP_KERNEL_MATERN52: KM_LAUNCH(GPIMHIP_KERNEL_MATERN52); break;
        case GPIMHIP_KERNEL_RQ: KM_LAUNCH(GPIMHIP_KERNEL_RQ); break;
        default: gpim_set_error("unknown kernel kind"); return GPIMHIP_E_BADARG;
    }
#undef KM_LAUNCH
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// small utilities
// ------------------------------------------------------------------------------------------
__global__ void pad_copy_kernel(const double* __restrict__ src, int64_t n, double* __restrict__ dst, int64_t np) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    src += blockIdx.y * n;
    dst += blockIdx.y * np;
    if (i < np) dst[i] = (i < n) ? src[i] : 0.0;
}
int launch_pad_copy(gpimhip_ctx* h, const double* src, int64_t n, double* dst, int64_t np) {
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((np + 255) / 256), h->nbatch), dim3(256), 0, h->stream, src,
                       n, dst, np);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// A[k,k] <- dinv[k] for every diagonal block (leaves of the triangular inversion)
template <typename R>
__global__ __launch_bounds__(256) void diag_inv_copy_kernel(R* __restrict__ A, int64_t ld,
                                                            const R* __restrict__ dinv, int64_t a_bs,
                                                            int64_t d_bs) {
    typedef typename Vec2<R>::T R2;
    const int k = blockIdx.x;
    A += blockIdx.y * a_bs;
    dinv += blockIdx.y * d_bs;
    R* dst = A + ((int64_t)k * NB) * ld + (int64_t)k * NB;
    const R* src = dinv + (int64_t)k * NB * NB;
    for (int e = threadIdx.x; e < NB * NB / 2; e += 256) {
        const int r = e >> 6, c2 = (e & 63) * 2;
        *reinterpret_cast<R2*>(dst + (int64_t)r * ld + c2) = *reinterpret_cast<const R2*>(src + r * NB + c2);
    }
}
int launch_diag_inv_copy(gpimhip_ctx* h, double* A, int64_t ld, int nb) {
    if (h->fp32)
        hipLaunchKernelGGL(diag_inv_copy_kernel<float>, dim3(nb, h->nbatch), dim3(256), 0, h->stream,
                           reinterpret_cast<float*>(A), ld, reinterpret_cast<const float*>(h->dinv), (int64_t)nb * NB * ld,
                           (int64_t)nb * NB * NB);
    else
        hipLaunchKernelGGL(diag_inv_copy_kernel<double>, dim3(nb, h->nbatch), dim3(256), 0, h->stream, A, ld, h->dinv,
                       (int64_t)nb * NB * ld, (int64_t)nb * NB * NB);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// copy an n x n user matrix into / out of the padded workspace (identity padding)
__global__ void pad_matrix_in_kernel(const double* __restrict__ src, int64_t n, int64_t lds_,
                                     double* __restrict__ dst, int64_t np) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= np * np) return;
    const int64_t r = idx / np, c = idx % np;
    double v;
    if (r < n && c < n) v = src[r * lds_ + c];
    else v = (r == c) ? 1.0 : 0.0;
    dst[idx] = v;
}
__global__ void pad_matrix_out_lower_kernel(const double* __restrict__ src, int64_t np,
                                            double* __restrict__ dst, int64_t n, int64_t ldd) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t r = idx / n, c = idx % n;
    if (c <= r) dst[r * ldd + c] = src[r * np + c];
}
int launch_pad_matrix_in(gpimhip_ctx* h, const double* src, int64_t n, int64_t ld, double* dst, int64_t np) {
    hipLaunchKernelGGL(pad_matrix_in_kernel, dim3((unsigned)((np * np + 255) / 256)), dim3(256), 0, h->stream,
                       src, n, ld, dst, np);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
int launch_pad_matrix_out_lower(gpimhip_ctx* h, const double* src, int64_t np, double* dst, int64_t n, int64_t ld) {
    hipLaunchKernelGGL(pad_matrix_out_lower_kernel, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0,
                       h->stream, src, np, dst, n, ld);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// triangular matrix-vector products with L^-1 (lower, zeros above the diagonal inside blocks)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// z[i] = sum_{j < jend(i)} L[i][j] * y[j];  one wave per row, 4 rows per workgroup
template <typename R>
__global__ __launch_bounds__(256) void trmv_lower_kernel(const R* __restrict__ L, int64_t ld, int64_t np,
                                                         const double* __restrict__ y, double* __restrict__ z) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    L += blockIdx.y * np * ld;
    y += blockIdx.y * np;
    z += blockIdx.y * np;
    if (i >= np) return;
    const int64_t jend = (i / NB + 1) * NB;
    const R* row = L + i * ld;
    double s = 0.0;
    for (int64_t j = lane * 2; j < jend; j += 128) {
        const typename Vec2<R>::T l = *reinterpret_cast<const typename Vec2<R>::T*>(row + j);
        const d2 v = *reinterpret_cast<const d2*>(y + j);
        s = fma((double)l[0], v[0], s);
        s = fma((double)l[1], v[1], s);
    }
    s = wave_sum(s);
    if (lane == 0) z[i] = s;
}

// out[j] = sum_{i >= i0(j)} A[i][j] * x[i] for 64 columns per workgroup; tri != 0 starts at the
// diagonal block of the column (A lower triangular), else at row 0.  HBM-bound (8 B per entry of
// A): 16 waves per workgroup, each with four 512-byte row segments in flight; fixed summation order.
#define GEMVT_WAVES 16
// (one workgroup per 64-column strip; the row-chunked form that alpha = L^-T z used until round 5 is gemv_t_tri_kernel below)
template <typename R>
__global__ __launch_bounds__(GEMVT_WAVES * 64) void gemv_t_kernel(const R* __restrict__ A, int64_t ld,
                                                                 int64_t nrows, const double* __restrict__ x,
                                                                 double* __restrict__ out, int tri, int64_t a_bs,
                                                                 int64_t x_bs, int64_t o_bs) {
    __shared__ double red[GEMVT_WAVES][64];
    A += blockIdx.y * a_bs;
    x += blockIdx.y * x_bs;
    out += blockIdx.y * o_bs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = (int64_t)blockIdx.x * 64 + lane;
    const int64_t i0 = tri ? ((int64_t)blockIdx.x * 64 / NB) * NB : 0, i1 = nrows;
    const R* col = A + j;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = i0 + wave;
    for (; i + 3 * GEMVT_WAVES < i1; i += 4 * GEMVT_WAVES) {
        const double a0 = (double)col[i * ld], a1 = (double)col[(i + GEMVT_WAVES) * ld];
        const double a2 = (double)col[(i + 2 * GEMVT_WAVES) * ld], a3 = (double)col[(i + 3 * GEMVT_WAVES) * ld];
        s0 = fma(a0, x[i], s0);
        s1 = fma(a1, x[i + GEMVT_WAVES], s1);
        s2 = fma(a2, x[i + 2 * GEMVT_WAVES], s2);
        s3 = fma(a3, x[i + 3 * GEMVT_WAVES], s3);
    }
    for (; i < i1; i += GEMVT_WAVES) s0 = fma((double)col[i * ld], x[i], s0);
    red[wave][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (wave == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < GEMVT_WAVES; ++w) t += red[w][lane];
        out[j] = t;
    }
}

int launch_trmv_lower(gpimhip_ctx* h, const double* L, int64_t ld, int64_t np, const double* y, double* z) {
    if (h->fp32)
        hipLaunchKernelGGL(trmv_lower_kernel<float>, dim3((unsigned)((np + 3) / 4), h->nbatch), dim3(256), 0, h->stream,
                           reinterpret_cast<const float*>(L), ld, np, y, z);
    else
        hipLaunchKernelGGL(trmv_lower_kernel<double>, dim3((unsigned)((np + 3) / 4), h->nbatch), dim3(256), 0, h->stream, L, ld,
                       np, y, z);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
int launch_gemv_t(gpimhip_ctx* h, const double* A, int64_t ld, int64_t nrows, int64_t ncols, const double* x,
                  double* out, int tri, int64_t a_bs, int64_t x_bs, int64_t o_bs) {
    const dim3 grid((unsigned)(ncols / 64), h->nbatch);
    if (h->fp32)
        hipLaunchKernelGGL(gemv_t_kernel<float>, grid, dim3(GEMVT_WAVES * 64), 0, h->stream,
                           reinterpret_cast<const float*>(A), ld, nrows, x, out, tri, a_bs, x_bs, o_bs);
    else
        hipLaunchKernelGGL(gemv_t_kernel<double>, grid, dim3(GEMVT_WAVES * 64), 0, h->stream, A, ld, nrows,
                       x, out, tri, a_bs, x_bs, o_bs);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// alpha = L^-T z over the lower-triangular L^-1 (round 6).  The strip of a 64-column block starts at its diagonal
// block, so strips run from np rows down to 128: cutting every strip into the SAME number of chunks (round 5's form of the kernel above)
// leaves the launch waiting for the first strips' workgroups -- 1.5 TB/s at np = 16384.  Here the ROWS are cut into
// chunks of rc (gemv_tri_rc: ~np / 32, a multiple of 128, 128 ... 1024): workgroup (column block, chunk) handles what
// lies at or below the block's diagonal block -- equal work per workgroup, thousands of workgroups of four waves --
// and writes 64 partial sums to part[chunk][column].  A wave reads two rows x 512 bytes per instruction (lane: row
// parity, column pair), four instructions in flight.  alpha_j = sum over the chunks r >= r0(j) in increasing order:
// gemv_tri_sum_kernel, or the consumer itself (grad_reduce_kernel sums the 256 entries it needs: gemv_tri_alpha).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double gemv_tri_alpha(const double* __restrict__ part, int nR, int rc, int64_t np, int64_t j) {
    double t = 0.0;
    for (int r = (int)(((j / NB) * NB) / rc); r < nR; ++r) t += part[(int64_t)r * np + j];
    return t;
}
template <typename R>
__global__ __launch_bounds__(256) void gemv_t_tri_kernel(const R* __restrict__ A, int64_t ld, int64_t np,
                                                         const double* __restrict__ x, double* __restrict__ part, int rc,
                                                         int nR, int64_t a_bs) {
    __shared__ d2 red[4][32];
    const int cb = blockIdx.x / nR, r = blockIdx.x % nR;
    const int64_t diag0 = ((int64_t)cb * 64 / NB) * NB;
    int64_t i0 = (int64_t)r * rc, i1 = i0 + rc < np ? i0 + rc : np;
    if (i1 <= diag0) return;                                   // structurally zero: never read by the sum
    if (i0 < diag0) i0 = diag0;
    A += blockIdx.y * a_bs;
    x += blockIdx.y * np;
    part += ((int64_t)blockIdx.y * nR + r) * np;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cp = lane & 31, rp = lane >> 5;
    typedef typename Vec2<R>::T RV;
    const R* col = A + (int64_t)cb * 64 + 2 * cp;
    d2 s0 = (d2){0.0, 0.0}, s1 = s0, s2 = s0, s3 = s0;
    int64_t i = i0 + 2 * wave + rp;                             // rows i, i + 8, i + 16, ... (8 = 4 waves x 2 rows)
    for (; i + 24 < i1; i += 32) {
        const RV a0 = *reinterpret_cast<const RV*>(col + i * ld), a1 = *reinterpret_cast<const RV*>(col + (i + 8) * ld);
        const RV a2 = *reinterpret_cast<const RV*>(col + (i + 16) * ld), a3 = *reinterpret_cast<const RV*>(col + (i + 24) * ld);
        const double x0 = x[i], x1 = x[i + 8], x2 = x[i + 16], x3 = x[i + 24];
        s0[0] = fma((double)a0[0], x0, s0[0]); s0[1] = fma((double)a0[1], x0, s0[1]);
        s1[0] = fma((double)a1[0], x1, s1[0]); s1[1] = fma((double)a1[1], x1, s1[1]);
        s2[0] = fma((double)a2[0], x2, s2[0]); s2[1] = fma((double)a2[1], x2, s2[1]);
        s3[0] = fma((double)a3[0], x3, s3[0]); s3[1] = fma((double)a3[1], x3, s3[1]);
    }
    for (; i < i1; i += 8) {
        const RV a0 = *reinterpret_cast<const RV*>(col + i * ld);
        const double x0 = x[i];
        s0[0] = fma((double)a0[0], x0, s0[0]); s0[1] = fma((double)a0[1], x0, s0[1]);
    }
    d2 v;
    v[0] = (s0[0] + s1[0]) + (s2[0] + s3[0]);
    v[1] = (s0[1] + s1[1]) + (s2[1] + s3[1]);
    v[0] += __shfl_xor(v[0], 32);                               // the two row parities (both halves end with the same sum)
    v[1] += __shfl_xor(v[1], 32);
    if (rp == 0) red[wave][cp] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        d2 t = red[0][cp];
        t[0] = ((t[0] + red[1][cp][0]) + red[2][cp][0]) + red[3][cp][0];
        t[1] = ((t[1] + red[1][cp][1]) + red[2][cp][1]) + red[3][cp][1];
        *reinterpret_cast<d2*>(part + (int64_t)cb * 64 + 2 * cp) = t;
    }
}
__global__ __launch_bounds__(256) void gemv_tri_sum_kernel(const double* __restrict__ part, int nR, int rc, int64_t np,
                                                           double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= np) return;
    out[blockIdx.y * np + j] = gemv_tri_alpha(part + (int64_t)blockIdx.y * nR * np, nR, rc, np, j);
}
// part: gemv_tri_chunks(np) * np doubles per problem (h->gemv_part); out == nullptr: the consumer sums the partials itself
int launch_gemv_t_tri(gpimhip_ctx* h, const double* A, int64_t ld, int64_t np, const double* x, double* part, double* out) {
    const int rc = gemv_tri_rc(np), nR = gemv_tri_chunks(np);
    const dim3 grid((unsigned)((np / 64) * nR), h->nbatch);
    if (h->fp32)
        hipLaunchKernelGGL(gemv_t_tri_kernel<float>, grid, dim3(256), 0, h->stream, reinterpret_cast<const float*>(A), ld, np, x,
                           part, rc, nR, np * ld);
    else
        hipLaunchKernelGGL(gemv_t_tri_kernel<double>, grid, dim3(256), 0, h->stream, A, ld, np, x, part, rc, nR, np * ld);
    HIP_TRY(hipGetLastError());
    if (out) {
        hipLaunchKernelGGL(gemv_tri_sum_kernel, dim3((unsigned)((np + 255) / 256), h->nbatch), dim3(256), 0, h->stream,
                           (const double*)part, nR, rc, np, out);
        HIP_TRY(hipGetLastError());
    }
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// residual r = y - (K + (jitter + noise) I) alpha with K generated on the fly in double (kmat_kernel's
// arithmetic): the iterative-refinement step of single-precision handles.  alpha = K^-1 y through an fp32 factor
// and its explicit fp32 inverse is off by eps32 * cond(K); one pass  alpha += K32^-1 (y - K alpha)  squares that
// factor.  Workgroup = 128 rows x every S-th column block; partial sums in part[s][row], summed in order.
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void kres_partial_kernel(const double* __restrict__ X, int64_t N, int d,
                                                           const double* __restrict__ alpha,
                                                           const ThetaDev* __restrict__ th, double* __restrict__ part,
                                                           int64_t x_bs, int64_t np, int S) {
    __shared__ double xz[128][5];
    __shared__ double al[128];
    __shared__ double red[128];
    const int tid = threadIdx.x, r = tid & 127, half = tid >> 7;
    const int ci = blockIdx.x / S, s0 = blockIdx.x % S, nb = (int)(np / NB);
    X += blockIdx.y * x_bs;
    alpha += blockIdx.y * np;
    th += blockIdx.y;
    part += (int64_t)blockIdx.y * S * np;
    const ThetaDev t = *th;
    const int64_t gi = (int64_t)ci * 128 + r;
    double a[4], an = 0.0;
#pragma unroll
    for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
        a[k] = (k < d && gi < N) ? X[gi * d + k] / t.ls[k] : 0.0;
        an += a[k] * a[k];
    }
    double acc = 0.0;
    for (int cj = s0; cj < nb; cj += S) {
        __syncthreads();
        if (tid < 128) {
            const int64_t g = (int64_t)cj * 128 + tid;
            double s2 = 0.0;
            for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
                const double v = (k < d && g < N) ? X[g * d + k] / t.ls[k] : 0.0;
                xz[tid][k] = v;
                s2 += v * v;
            }
            xz[tid][4] = s2;
            al[tid] = (g < N) ? alpha[g] : 0.0;
        }
        __syncthreads();
        for (int c = half * 64; c < half * 64 + 64; ++c) {
            double dot = a[0] * xz[c][0];
            dot = fma(a[1], xz[c][1], dot);
            dot = fma(a[2], xz[c][2], dot);
            dot = fma(a[3], xz[c][3], dot);
            const double r2 = clamp0_nan((an - 2.0 * dot) + xz[c][4]);
            acc = fma(t.var * kfun_value<KIND>(r2, t.alpha), al[c], acc);
        }
    }
    __syncthreads();
    if (half == 1) red[r] = acc;
    __syncthreads();
    if (half == 0) part[(int64_t)s0 * np + gi] = (gi < N) ? acc + red[r] : 0.0;
}
__global__ void kres_final_kernel(const double* __restrict__ y, const double* __restrict__ alpha,
                                  const double* __restrict__ part, const ThetaDev* __restrict__ th, int64_t N,
                                  int64_t np, int S, double* __restrict__ res) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    const int64_t o = (int64_t)blockIdx.y * np;
    double s = 0.0;
    for (int q = 0; q < S; ++q) s += part[((int64_t)blockIdx.y * S + q) * np + i];
    res[o + i] = (i < N) ? (y[o + i] - th[blockIdx.y].diag_add * alpha[o + i]) - s : 0.0;
}
__global__ void axpy_kernel(double* __restrict__ x, const double* __restrict__ d, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += d[i];
}
// res <- ypad - K(theta) alpha   (scratch: S * np doubles per problem)
int launch_kres(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, int64_t N, double* scratch,
                int S, double* res) {
    const int64_t np = h->np;
    const int nb = (int)(np / NB);
    dim3 grid(nb * S, h->nbatch), block(256);
#define KR_LAUNCH(KIND) \
    hipLaunchKernelGGL((kres_partial_kernel<KIND>), grid, block, 0, h->stream, X, N, m->dim, h->alpha, h->theta, scratch, x_bs, np, S)
    switch (m->kernel) {
        case GPIMHIP_KERNEL_RBF: KR_LAUNCH(GPIMHIP_KERNEL_RBF); break;
        case GPIMHIP_KERNEL_MATERN52: KR_LAUNCH(GPIMHIP_KERNEL_MATERN52); break;
        case GPIMHIP_KERNEL_RQ: KR_LAUNCH(GPIMHIP_KERNEL_RQ); break;
        default: gpim_set_error("unknown kernel kind"); return GPIMHIP_E_BADARG;
    }
#undef KR_LAUNCH
    hipLaunchKernelGGL(kres_final_kernel, dim3((unsigned)((np + 255) / 256), h->nbatch), dim3(256), 0, h->stream, h->ypad,
                       h->alpha, scratch, h->theta, N, np, S, res);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
int launch_axpy(gpimhip_ctx* h, double* x, const double* d, int64_t n) {
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, x, d, n);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// finalize: sums the partials in a fixed tree, forms loss and d loss/du, then (fit mode) applies one torch.optim.Adam
// step to u and records the constrained values.  Run by ONE workgroup of 256 threads: finalize_kernel, or (round 6) the
// LAST workgroup of grad_reduce_kernel to finish (FinFused) -- the Adam step is then not a launch of its own, and the
// theta of the stepped parameters is written for the next iteration's kmat_kernel (no theta launch either).
// ------------------------------------------------------------------------------------------
__device__ double block_sum_256(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
// The same tree for NQ quantities at once with three barriers: red[t] += red[t + 128], += red[t + 64] in LDS order, then
// lane t += lane t + s for s = 32 ... 1 inside wave 0 -- what block_sum_256's steps compute for t < s, bit for bit.
// arr: NQ x 256 doubles of LDS.  Results in out[0 .. NQ) (LDS), valid after the call for every thread.
template <int NQ>
__device__ __forceinline__ void block_sum_256_multi(const double* v, double* arr, double* out) {
    const int tid = threadIdx.x;
    __syncthreads();                                   // (arr may alias what the caller still read)
#pragma unroll
    for (int q = 0; q < NQ; ++q) arr[q * 256 + tid] = v[q];
    __syncthreads();
    if (tid < 64) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double* a = arr + q * 256;
            double x = (a[tid] + a[tid + 128]) + (a[tid + 64] + a[tid + 192]);
            x += __shfl_down(x, 32);
            x += __shfl_down(x, 16);
            x += __shfl_down(x, 8);
            x += __shfl_down(x, 4);
            x += __shfl_down(x, 2);
            x += __shfl_down(x, 1);
            if (tid == 0) out[q] = x;
        }
    }
    __syncthreads();
}

// Iteration-indexed form: when `iter` is given the Adam bias corrections, the history row and the
// loss slot are looked up by the device-resident iteration counter, which this kernel advances.  All
// T iterations then enqueue byte-identical launches, i.e. one captured hipGraph can be replayed.
struct FinalizeIter {
    int32_t* iter;              // device counter (null: use the by-value arguments below)
    const double* bc;           // [2*T]: lr/(1-beta1^t) then sqrt(1-beta2^t)
    int32_t T;
    double* hist_base;          // T x P or null
    double* loss_base;          // T or null
};

// everything the finalize step of one problem needs (pointers already offset to the problem by the caller)
struct FinProblem {
    const double* grad_part; const double* z; const double* zb; const double* logdet_part;
    ThetaDev* th; double* u; double* adam_m; double* adam_v;
    FinalizeIter fi;
};
// COHERENT: the partial sums were written by other workgroups of the SAME launch (device-scope loads)
template <bool COHERENT>
__device__ __forceinline__ double fin_load(const double* p) {
    return COHERENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
// arr: 5 x 256 doubles of LDS, S: 9 doubles of LDS.  theta_next != nullptr: the theta of the stepped parameters goes there.
template <bool COHERENT>
__device__ __forceinline__ void finalize_block(const gpimhip_model_t& m, int64_t N, int64_t np, int nb, int ntile, FinProblem f,
                                               int do_adam, AdamStep st, double* loss_out, double* grad_out, double* hist_row,
                                               int32_t* info, ThetaDev* theta_next, double* arr, double* S) {
    const int tid = threadIdx.x;
    double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    // (four tiles' loads in flight, added in the order q = tid, tid + 256, ...: device-scope loads are round trips to
    // memory, and one tile per trip made this loop 130 us at N = 16384)
    for (int q0 = tid; q0 < ntile; q0 += 4 * 256) {
        double t[4][7];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = q0 + 256 * i;
            const double* g = f.grad_part + (int64_t)(q < ntile ? q : q0) * 8;
#pragma unroll
            for (int k = 0; k < 7; ++k) t[i][k] = fin_load<COHERENT>(g + k);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q0 + 256 * i < ntile) {
#pragma unroll
                for (int k = 0; k < 7; ++k) v[k] += t[i][k];
            }
    }
    for (int64_t i = tid; i < np; i += 256) v[7] = fma(f.z[i], f.zb[i], v[7]);     // |L^-1 y|^2, or y^T alpha (refined, fp32 matrices)
    for (int k = tid; k < nb; k += 256) v[8] += f.logdet_part[k];
    block_sum_256_multi<5>(v, arr, S);
    block_sum_256_multi<4>(v + 5, arr, S + 5);
    if (tid != 0) return;
    if (f.fi.iter) {
        const int it = *f.fi.iter;
        // A factorisation of this training loop has failed (this iteration's, or an earlier one of any
        // problem of the batch): the reference raises here (gpr.py:192), with u, the Adam state and the
        // history as the previous iteration left them.  Freeze them; record how far the loop got.
        if (*info != 0) {
            atomicMin(info + 1, it);
            return;
        }
        const int P = 2 + m.n_ls + (m.kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
        st.lr_over_bc1 = f.fi.bc[it];
        st.bc2_sqrt = f.fi.bc[f.fi.T + it];
        loss_out = f.fi.loss_base ? f.fi.loss_base + it : nullptr;
        hist_row = f.fi.hist_base ? f.fi.hist_base + (int64_t)it * P : nullptr;
        *f.fi.iter = it + 1;
    }
    // working storage in LDS (arr is free again): gradient, the current theta (theta_next may be f.th itself), the next one
    double* g = arr;
    ThetaDev* told = reinterpret_cast<ThetaDev*>(arr + 16);
    ThetaDev* tn = reinterpret_cast<ThetaDev*>(arr + 48);
    *told = *f.th;
    finalize_step_ws(m, N, S, S[7], S[8], *told, f.u, f.adam_m, f.adam_v, do_adam, st, loss_out, grad_out, hist_row,
                     prior_constant(m), theta_next, g, *tn);
}
__device__ __forceinline__ FinProblem fin_problem(const gpimhip_model_t& m, int64_t b, int64_t np, int nb, int ntile,
                                                  const double* grad_part, const double* z, const double* zb,
                                                  const double* logdet_part, ThetaDev* th, double* u, double* adam_m,
                                                  double* adam_v, FinalizeIter fi) {
    const int P = 2 + m.n_ls + (m.kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    FinProblem f;
    f.grad_part = grad_part + b * ntile * 8;
    f.z = z + b * np;
    f.zb = zb + b * np;
    f.logdet_part = logdet_part + b * nb;
    f.th = th + b;
    f.u = u + b * P;
    f.adam_m = adam_m + b * MAXP;
    f.adam_v = adam_v + b * MAXP;
    f.fi = fi;
    if (fi.iter) {
        f.fi.iter += b;
        if (fi.hist_base) f.fi.hist_base += b * fi.T * P;
        if (fi.loss_base) f.fi.loss_base += b * fi.T;
    }
    return f;
}

__global__ __launch_bounds__(256) void finalize_kernel(gpimhip_model_t m, int64_t N, int64_t np, int nb,
                                                       int ntile, const double* __restrict__ grad_part,
                                                       const double* __restrict__ z, const double* __restrict__ zb,
                                                       const double* __restrict__ logdet_part,
                                                       ThetaDev* __restrict__ th, double* __restrict__ u,
                                                       double* __restrict__ adam_m, double* __restrict__ adam_v,
                                                       int do_adam, AdamStep st, double* __restrict__ loss_out,
                                                       double* __restrict__ grad_out,
                                                       double* __restrict__ hist_row, FinalizeIter fi,
                                                       int32_t* __restrict__ info, int carry_theta) {
    __shared__ double arr[5 * 256];
    __shared__ double S[9];
    const FinProblem f = fin_problem(m, blockIdx.y, np, nb, ntile, grad_part, z, zb, logdet_part, th, u, adam_m, adam_v, fi);
    finalize_block<false>(m, N, np, nb, ntile, f, do_adam, st, loss_out, grad_out, hist_row, info,
                          carry_theta ? th + blockIdx.y : nullptr, arr, S);
}

// the finalize step as the tail of grad_reduce_kernel
struct FinFused {
    int enabled;
    int do_adam;
    int carry_theta;            // write the theta of the stepped parameters over the current one
    int nb;
    uint32_t* counter;          // per problem, 0 between launches
    gpimhip_model_t m;
    int64_t N;
    const double* z; const double* zb; const double* logdet_part;
    double* u; double* adam_m; double* adam_v;
    AdamStep st;
    double* loss_out; double* grad_out; double* hist_row;
    FinalizeIter fi;
    int32_t* info;
};

// ------------------------------------------------------------------------------------------
// gradient reduction over the lower tiles of K^-1:
//   w_ij = (2 - delta_ij) * (Kinv_ij - alpha_i alpha_j)
//   S[0]      += w * k/s2                       -> d/d s2
//   S[1+k]    += w * s2*h * (a_ik - a_jk)^2     -> d/d l_k  (times 1/l_k later)
//   S[5]      += delta_ij * (Kinv_ii - alpha_i^2) -> d/d noise
//   S[6]      += w * s2 * dk/dalpha / s2        -> d/d alpha (RQ)
// ------------------------------------------------------------------------------------------
// alpha_part != nullptr: alpha is given as the row-chunk partial sums of gemv_t_tri_kernel (nR chunks of rc rows) and
// every workgroup adds up the 256 entries it needs itself (no sum launch).  fin.enabled: the last workgroup of a problem to
// finish runs the finalize step (FinFused).
template <int KIND, typename R>
__global__ __launch_bounds__(256) void grad_reduce_kernel(const R* __restrict__ Kinv, int64_t ld,
                                                          const double* __restrict__ X, int64_t N, int d,
                                                          const double* __restrict__ alpha,
                                                          ThetaDev* __restrict__ th,
                                                          double* __restrict__ part, int64_t x_bs, int64_t np,
                                                          const TileDesc* __restrict__ tiles,
                                                          const double* __restrict__ alpha_part, int nR, int rc, FinFused fin) {
    // one buffer, carved: the staging of the tile phase, then (last workgroup only) the reduction scratch of the finalize step
    __shared__ double sh[2 * 128 * 5 + 2 * 128 + 4 * 8 + 16];
    double (*xa)[5] = reinterpret_cast<double (*)[5]>(sh);
    double (*xz)[5] = reinterpret_cast<double (*)[5]>(sh + 128 * 5);
    double* al_r = sh + 2 * 128 * 5;
    double* al_c = al_r + 128;
    double (*red)[8] = reinterpret_cast<double (*)[8]>(al_c + 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* part_all = part;
    Kinv += blockIdx.y * np * ld;
    X += blockIdx.y * x_bs;
    if (alpha) alpha += blockIdx.y * np;
    if (alpha_part) alpha_part += (int64_t)blockIdx.y * nR * np;
    part += (int64_t)blockIdx.y * gridDim.x * 8;
    // tiles == nullptr: workgroup q is lower tile q of a full np x np matrix.  Otherwise (distributed layouts:
    // a rank holds some block columns side by side) tile (ci, cj) of the matrix lives at block column kb0 of Kinv.
    int ci, cj;
    int64_t ccol;
    if (tiles) {
        const TileDesc td = tiles[blockIdx.x];
        ci = td.ci; cj = td.cj; ccol = (int64_t)td.kb0 * 128;
    } else {
        lower_tile_from_linear(blockIdx.x, ci, cj);
        ccol = (int64_t)cj * 128;
    }
    const ThetaDev t = th[blockIdx.y];
    {
        const bool isrow = tid < 128;
        const int loc = tid & 127;
        const int64_t g = (int64_t)(isrow ? ci : cj) * 128 + loc;
        double s2 = 0.0;
        double (*dst)[5] = isrow ? xa : xz;
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
            double a = 0.0;
            if (k < d && g < N) a = X[g * d + k] / t.ls[k];
            dst[loc][k] = a;
            s2 += a * a;
        }
        dst[loc][4] = s2;
        double av = 0.0;
        if (g < N) av = alpha_part ? gemv_tri_alpha(alpha_part, nR, rc, np, g) : alpha[g];
        (isrow ? al_r : al_c)[loc] = av;
    }
    __syncthreads();
    double S[7] = {0, 0, 0, 0, 0, 0, 0};
    const int ty = tid >> 4, tx = tid & 15;
    // strictly lower tiles inside the matrix: every entry counts twice, none is skipped (no per-entry 64-bit compares)
    const bool interior = ci > cj && (int64_t)ci * 128 + 128 <= N;
    auto body = [&](auto INTERIOR) {
    for (int rr = 0; rr < 8; ++rr) {
        const int r = ty + 16 * rr;
        const int64_t gi = (int64_t)ci * 128 + r;
        const double a0 = xa[r][0], a1 = xa[r][1], a2 = xa[r][2], a3 = xa[r][3], an = xa[r][4];
        const double ali = al_r[r];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const typename Vec2<R>::T kv = *reinterpret_cast<const typename Vec2<R>::T*>(Kinv + gi * ld + ccol + tx * 2 + 32 * cc);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = tx * 2 + 32 * cc + e;
                const int64_t gj = (int64_t)cj * 128 + c;
                if (!decltype(INTERIOR)::value && (gi >= N || gj > gi)) continue;
                const double g = (double)kv[e] - ali * al_c[c];
                const double w = (!decltype(INTERIOR)::value && gi == gj) ? g : 2.0 * g;
                double dot = a0 * xz[c][0];
                dot = fma(a1, xz[c][1], dot);
                dot = fma(a2, xz[c][2], dot);
                dot = fma(a3, xz[c][3], dot);
                double r2 = clamp0_nan((an - 2.0 * dot) + xz[c][4]);
                const KVal kvv = kfun_grad<KIND>(r2, t.alpha);
                S[0] = fma(w, kvv.e, S[0]);
                const double wh = w * kvv.h;
                const double d0 = a0 - xz[c][0], d1 = a1 - xz[c][1], d2_ = a2 - xz[c][2], d3 = a3 - xz[c][3];
                S[1] = fma(wh, d0 * d0, S[1]);
                S[2] = fma(wh, d1 * d1, S[2]);
                S[3] = fma(wh, d2_ * d2_, S[3]);
                S[4] = fma(wh, d3 * d3, S[4]);
                if (!decltype(INTERIOR)::value && gi == gj) S[5] += g;
                if (KIND == GPIMHIP_KERNEL_RQ) S[6] = fma(w, kvv.ga, S[6]);
            }
        }
    }
    };
    if (interior) body(std::true_type{});
    else body(std::false_type{});
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double v = wave_sum(S[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < 8) {
        double v = 0.0;
        if (tid < 7) v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        // fused finalize: a device-scope (write-through) store -- the reader may sit on another XCD, whose L2 is not
        // coherent with this one.  A release FENCE here instead (__threadfence: write back + invalidate the whole L2, once
        // per workgroup) was measured at 1.87 ms for this launch at N = 16384 (0.40 without).
        if (fin.enabled) __hip_atomic_store(part + (int64_t)blockIdx.x * 8 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else part[(int64_t)blockIdx.x * 8 + tid] = v;
    }
    if (!fin.enabled) return;
    // ---- the last workgroup of this problem runs the finalize step (loss, gradient, Adam, next theta)
    int* flag = reinterpret_cast<int*>(sh + 2 * 128 * 5 + 2 * 128 + 4 * 8);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // s_waitcnt: this workgroup's partial sums have been written ...
    __syncthreads();
    if (tid == 0) {
        // ... before it counts as finished (one counter per problem; a two-level count -- groups of 32 workgroups -- measured the
        // same at 8256 workgroups: the adds are spread over the launch)
        const unsigned int done = atomicAdd(fin.counter + blockIdx.y, 1u);
        *flag = (done == gridDim.x - 1);
        if (*flag) fin.counter[blockIdx.y] = 0;        // ready for the next launch (every other workgroup has passed)
    }
    __syncthreads();
    if (!*flag) return;
    // (the other workgroups' partial sums are read with device-scope loads: finalize_block<true>)
    const int ntile = (int)gridDim.x;
    const FinProblem f = fin_problem(fin.m, blockIdx.y, np, fin.nb, ntile, part_all, fin.z, fin.zb, fin.logdet_part, th, fin.u,
                                     fin.adam_m, fin.adam_v, fin.fi);
    // (sh: 1584 doubles; the reduction scratch takes the first 1280, the nine sums sit behind the flag)
    finalize_block<true>(fin.m, fin.N, np, fin.nb, ntile, f, fin.do_adam, fin.st, fin.loss_out, fin.grad_out, fin.hist_row,
                         fin.info, fin.carry_theta ? th + blockIdx.y : nullptr, sh, sh + 2 * 128 * 5 + 2 * 128 + 4 * 8 + 2);
}

// alpha_part: nullptr (alpha is a vector) or the row-chunk partial sums of launch_gemv_t_tri.  fin: nullptr, or the finalize
// step to run in the launch's last workgroup (launch_grad_reduce_fin).
static int launch_grad_reduce_impl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld,
                                   const double* X, int64_t N, int nb, const double* alpha, int64_t x_bs,
                                   const double* alpha_part, const FinFused* fin) {
    const int ntile = nb * (nb + 1) / 2;
    const int64_t np = (int64_t)nb * NB;
    dim3 grid(ntile, h->nbatch), block(256);
    FinFused ff;
    memset(&ff, 0, sizeof(ff));
    if (fin) ff = *fin;
    const int nR = alpha_part ? gemv_tri_chunks(np) : 0, rc = alpha_part ? gemv_tri_rc(np) : 0;
#define GR_LAUNCH(KIND)                                                                                   \
    do {                                                                                                  \
        if (h->fp32)                                                                                      \
            hipLaunchKernelGGL((grad_reduce_kernel<KIND, float>), grid, block, 0, h->stream, reinterpret_cast<const float*>(Kinv), \
                               ld, X, N, m->dim, alpha, h->theta, h->grad_part, x_bs, np, (const TileDesc*)nullptr, alpha_part, nR, rc, ff);  \
        else                                                                                              \
            hipLaunchKernelGGL((grad_reduce_kernel<KIND, double>), grid, block, 0, h->stream, Kinv, ld, X, N, m->dim, alpha, \
                               h->theta, h->grad_part, x_bs, np, (const TileDesc*)nullptr, alpha_part, nR, rc, ff);  \
    } while (0)
    switch (m->kernel) {
        case GPIMHIP_KERNEL_RBF: GR_LAUNCH(GPIMHIP_KERNEL_RBF); break;
        case GPIMHIP_KERNEL_MATERN52: GR_LAUNCH(GPIMHIP_KERNEL_MATERN52); break;
        case GPIMHIP_KERNEL_RQ: GR_LAUNCH(GPIMHIP_KERNEL_RQ); break;
        default: gpim_set_error("unknown kernel kind"); return GPIMHIP_E_BADARG;
    }
#undef GR_LAUNCH
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
int launch_grad_reduce(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld,
                       const double* X, int64_t N, int nb, const double* alpha, int64_t x_bs) {
    return launch_grad_reduce_impl(h, m, Kinv, ld, X, N, nb, alpha, x_bs, nullptr, nullptr);
}
// The gradient contraction AND the finalize step (loss, gradient, Adam step, history row, next theta) in one launch: the
// arguments of launch_finalize, run by the last workgroup of every problem.  alpha_part: see launch_grad_reduce_impl.
// carry_theta: the theta of the stepped parameters replaces h->theta (the next iteration skips its theta launch).
int launch_grad_reduce_fin(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld, const double* X,
                           int64_t N, int nb, const double* alpha, int64_t x_bs, const double* alpha_part, double* u,
                           int do_adam, AdamStep st, double* loss_out, double* grad_out, double* hist_row, int32_t* iter,
                           const double* bc, int T, double* hist_base, double* loss_base, int carry_theta) {
    FinFused ff;
    memset(&ff, 0, sizeof(ff));
    ff.enabled = 1;
    ff.do_adam = do_adam;
    ff.carry_theta = carry_theta;
    ff.nb = nb;
    ff.counter = h->fin_counter;
    ff.m = *m;
    ff.N = N;
    ff.z = h->fp32 ? h->ypad : h->z;
    ff.zb = h->fp32 ? h->alpha : h->z;
    ff.logdet_part = h->logdet_part;
    ff.u = u; ff.adam_m = h->adam_m; ff.adam_v = h->adam_v;
    ff.st = st;
    ff.loss_out = loss_out; ff.grad_out = grad_out; ff.hist_row = hist_row;
    ff.fi = FinalizeIter{iter, bc, T, hist_base, loss_base};
    ff.info = h->info;
    return launch_grad_reduce_impl(h, m, Kinv, ld, X, N, nb, alpha, x_bs, alpha_part, &ff);
}

// carry_theta: the theta of the stepped parameters replaces h->theta (the next iteration skips its theta launch)
int launch_finalize(gpimhip_ctx* h, const gpimhip_model_t* m, int64_t N, int64_t np, double* u, int do_adam,
                    AdamStep st, double* loss_out, double* grad_out, double* hist_row, int32_t* iter,
                    const double* bc, int T, double* hist_base, double* loss_base, int carry_theta) {
    const int nb = (int)(np / NB);
    FinalizeIter fi{iter, bc, T, hist_base, loss_base};
    hipLaunchKernelGGL(finalize_kernel, dim3(1, h->nbatch), dim3(256), 0, h->stream, *m, N, np, nb, nb * (nb + 1) / 2,
                       h->grad_part, h->fp32 ? h->ypad : h->z, h->fp32 ? h->alpha : h->z, h->logdet_part, h->theta, u, h->adam_m,
                       h->adam_v, do_adam, st,
                       loss_out, grad_out, hist_row, fi, h->info, carry_theta);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// prediction epilogue: var_j = clamp(s2 - sum_ci colpart[ci][j], 0) + noise
// ------------------------------------------------------------------------------------------
__global__ void predict_var_kernel(const double* __restrict__ colpart, int64_t ldp, int nb, int64_t m0,
                                   int64_t mcount, const ThetaDev* __restrict__ th, double* __restrict__ var_out,
                                   int64_t M) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    colpart += blockIdx.y * nb * ldp;
    th += blockIdx.y;
    var_out += blockIdx.y * M;
    if (j >= mcount) return;
    double q = 0.0;
    for (int ci = 0; ci < nb; ++ci) q += colpart[(int64_t)ci * ldp + j];
    const double v = clamp0_nan(th->var - q);
    var_out[m0 + j] = v + th->noise;
}
__global__ void copy_slice_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n,
                                  int64_t s_bs, int64_t d_bs) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) dst[blockIdx.y * d_bs + j] = src[blockIdx.y * s_bs + j];
}
int launch_predict_var(gpimhip_ctx* h, int64_t ldp, int nb, int64_t m0, int64_t mcount, double* var_out, int64_t M) {
    hipLaunchKernelGGL(predict_var_kernel, dim3((unsigned)((mcount + 255) / 256), h->nbatch), dim3(256), 0, h->stream,
                       h->colpart, ldp, nb, m0, mcount, h->theta, var_out, M);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
int launch_copy_slice(gpimhip_ctx* h, const double* src, double* dst, int64_t n, int64_t s_bs, int64_t d_bs) {
    hipLaunchKernelGGL(copy_slice_kernel, dim3((unsigned)((n + 255) / 256), h->nbatch), dim3(256), 0, h->stream, src,
                       dst, n, s_bs, d_bs);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// acquisition sweep (acqfunc.py:11-92).  Phi / phi as scipy.stats.norm: ndtr(z) = erfc(-z/sqrt2)/2.
// ------------------------------------------------------------------------------------------
__global__ void acq_kernel(int kind, const double* __restrict__ mean, const double* __restrict__ sd, int64_t M,
                           double p0, double p1, const double* __restrict__ mask, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double mu = mean[j], s = sd[j];
    double a;
    if (kind == GPIMHIP_ACQ_CB) {
        a = p0 * mu + p1 * s;
    } else {
        const double imp = mu - p0 - p1;
        const double zz = imp / s;
        const double cdf = 0.5 * erfc(-zz * 0.7071067811865476);
        if (kind == GPIMHIP_ACQ_EI) {
            const double pdf = exp(-0.5 * zz * zz) * 0.3989422804014327;
            a = imp * cdf + s * pdf;
        } else {
            a = cdf;
        }
    }
    if (mask) a = mask[j] * a;
    out[j] = a;
}
int launch_acq(gpimhip_ctx* h, int kind, const double* mean, const double* sd, int64_t M, double p0, double p1,
               const double* mask, double* out) {
    hipLaunchKernelGGL(acq_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, kind, mean, sd, M,
                       p0, p1, mask, out);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// nanmax over n values, single workgroup (n is a grid size: <= a few 1e5)
__global__ __launch_bounds__(1024) void nanmax_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ out) {
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    double best = -INFINITY;
    bool any = false;
    for (int64_t i = tid; i < n; i += 1024) {
        const double v = x[i];
        if (v == v) { best = fmax(best, v); any = true; }
    }
    red[tid] = any ? best : -INFINITY;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) {
        // all-NaN input: np.nanmax returns NaN (with a warning)
        out[0] = (red[0] == -INFINITY) ? __builtin_nan("") : red[0];
    }
}
int launch_nanmax(gpimhip_ctx* h, const double* x, int64_t n, double* out) {
    hipLaunchKernelGGL(nanmax_kernel, dim3(1), dim3(1024), 0, h->stream, x, n, out);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// descending top-k by repeated arg-max inside one workgroup.  Keys are order-preserving 64-bit
// images of the doubles; NaN ranks above +inf when keep_nan (np.argsort puts NaN last, the caller
// reverses), and is excluded otherwise.  Ties: larger flat index first (= reversed stable sort).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long order_key(double v, int keep_nan) {
    if (v != v) return keep_nan ? 0xFFFFFFFFFFFFFFFFull : 0ull;
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    b = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
    // reserve 0 for "excluded" and all-ones for NaN
    if (b == 0ull) b = 1ull;
    if (b == 0xFFFFFFFFFFFFFFFFull) b = 0xFFFFFFFFFFFFFFFEull;
    return b;
}
__global__ void topk_keys_kernel(const double* __restrict__ x, int64_t M, int keep_nan,
                                 unsigned long long* __restrict__ keys) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < M) keys[j] = order_key(x[j], keep_nan);
}
__global__ __launch_bounds__(1024) void topk_select_kernel(const double* __restrict__ x,
                                                           unsigned long long* __restrict__ keys, int64_t M, int k,
                                                           double* __restrict__ vals, int64_t* __restrict__ idx,
                                                           int64_t* __restrict__ count) {
    __shared__ unsigned long long rk[1024];
    __shared__ int64_t ri[1024];
    const int tid = threadIdx.x;
    int64_t found = 0;
    for (int r = 0; r < k; ++r) {
        unsigned long long bk = 0ull;
        int64_t bi = -1;
        for (int64_t j = tid; j < M; j += 1024) {
            const unsigned long long kk = keys[j];
            if (kk > bk || (kk == bk && kk != 0ull && j > bi)) { bk = kk; bi = j; }
        }
        rk[tid] = bk;
        ri[tid] = bi;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if (tid < s) {
                const unsigned long long ok = rk[tid + s];
                const int64_t oi = ri[tid + s];
                if (ok > rk[tid] || (ok == rk[tid] && oi > ri[tid])) { rk[tid] = ok; ri[tid] = oi; }
            }
            __syncthreads();
        }
        const unsigned long long wk = rk[0];
        const int64_t wi = ri[0];
        __syncthreads();
        if (wk == 0ull || wi < 0) break;
        if (tid == 0) {
            vals[r] = x[wi];
            idx[r] = wi;
            keys[wi] = 0ull;
        }
        found = r + 1;
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) {
        *count = found;
        for (int r = (int)found; r < k; ++r) { vals[r] = __builtin_nan(""); idx[r] = -1; }
    }
}
int launch_topk(gpimhip_ctx* h, const double* x, int64_t M, int k, int keep_nan, double* vals, int64_t* idx,
                int64_t* count) {
    hipLaunchKernelGGL(topk_keys_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, x, M, keep_nan,
                       h->keys);
    hipLaunchKernelGGL(topk_select_kernel, dim3(1), dim3(1024), 0, h->stream, x, h->keys, M, k, vals, idx, count);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}


// ------------------------------------------------------------------------------------------
// distributed training (api.hip: gpimhip_dist_grad_sums / gpimhip_dist_finalize): the gradient contraction over the
// tiles of K^-1 a rank holds, reduced to the seven sums S[] in a fixed order; after the all-reduce of S across
// the ranks the same finalize_step as the single-GPU path turns them into loss, gradient and Adam step.
// ------------------------------------------------------------------------------------------
int launch_grad_reduce_tiles(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld, const double* X,
                             int64_t N, int64_t np, const double* alpha, const TileDesc* tiles, int ntile, double* part) {
    dim3 grid(ntile, 1), block(256);
    FinFused ff0;
    memset(&ff0, 0, sizeof(ff0));
    switch (m->kernel) {
        case GPIMHIP_KERNEL_RBF:
            hipLaunchKernelGGL((grad_reduce_kernel<GPIMHIP_KERNEL_RBF, double>), grid, block, 0, h->stream, Kinv, ld, X, N, m->dim,
                               alpha, h->theta, part, (int64_t)0, np, tiles, (const double*)nullptr, 0, 0, ff0);
            break;
        case GPIMHIP_KERNEL_MATERN52:
            hipLaunchKernelGGL((grad_reduce_kernel<GPIMHIP_KERNEL_MATERN52, double>), grid, block, 0, h->stream, Kinv, ld, X, N,
                               m->dim, alpha, h->theta, part, (int64_t)0, np, tiles, (const double*)nullptr, 0, 0, ff0);
            break;
        case GPIMHIP_KERNEL_RQ:
            hipLaunchKernelGGL((grad_reduce_kernel<GPIMHIP_KERNEL_RQ, double>), grid, block, 0, h->stream, Kinv, ld, X, N, m->dim,
                               alpha, h->theta, part, (int64_t)0, np, tiles, (const double*)nullptr, 0, 0, ff0);
            break;
        default: gpim_set_error("unknown kernel kind"); return GPIMHIP_E_BADARG;
    }
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

__global__ __launch_bounds__(256) void sum7_kernel(const double* __restrict__ part, int ntile, double* __restrict__ S) {
    __shared__ double red[256];
    for (int k = 0; k < 7; ++k) {
        double v = 0.0;
        for (int q = threadIdx.x; q < ntile; q += 256) v += part[(int64_t)q * 8 + k];
        v = block_sum_256(v, red);
        if (threadIdx.x == 0) S[k] = v;
    }
    if (threadIdx.x == 0) S[7] = 0.0;
}
int launch_sum7(gpimhip_ctx* h, const double* part, int ntile, double* S) {
    hipLaunchKernelGGL(sum7_kernel, dim3(1), dim3(256), 0, h->stream, part, ntile, S);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

__global__ void dist_finalize_kernel(gpimhip_model_t m, int64_t N, const double* __restrict__ S, double q2, double lg,
                                     const ThetaDev* __restrict__ th, double* __restrict__ u, double* __restrict__ adam_m,
                                     double* __restrict__ adam_v, int do_adam, AdamStep st, double* __restrict__ loss_out,
                                     double* __restrict__ grad_out, double* __restrict__ hist_row) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double Sl[8];
    for (int k = 0; k < 8; ++k) Sl[k] = S[k];
    finalize_step(m, N, Sl, q2, lg, *th, u, adam_m, adam_v, do_adam, st, loss_out, grad_out, hist_row, prior_constant(m));
}
// ... with the scalars where the distributed driver leaves them -- device memory: red[0..7] = the all-reduced gradient
// sums, red[8] = sum log L_ii, red[9] = how many ranks met a non-positive pivot (then nothing is touched but the loss,
// which becomes NaN: the driver raises after its one read-back of the iteration); quad[0] = y^T alpha
__global__ void dist_finalize_dev_kernel(gpimhip_model_t m, int64_t N, const double* __restrict__ red,
                                         const double* __restrict__ quad, const ThetaDev* __restrict__ th,
                                         double* __restrict__ u, double* __restrict__ adam_m, double* __restrict__ adam_v,
                                         int do_adam, AdamStep st, double* __restrict__ loss_out,
                                         double* __restrict__ grad_out, double* __restrict__ hist_row) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (red[9] != 0.0) {
        if (loss_out) *loss_out = __builtin_nan("");
        return;
    }
    double Sl[8];
    for (int k = 0; k < 8; ++k) Sl[k] = red[k];
    finalize_step(m, N, Sl, quad[0], red[8], *th, u, adam_m, adam_v, do_adam, st, loss_out, grad_out, hist_row, prior_constant(m));
}
int launch_dist_finalize_dev(gpimhip_ctx* h, const gpimhip_model_t* m, int64_t N, const double* red, const double* quad,
                             double* u, int do_adam, AdamStep st, double* loss_out, double* grad_out, double* hist_row) {
    hipLaunchKernelGGL(dist_finalize_dev_kernel, dim3(1), dim3(64), 0, h->stream, *m, N, red, quad, h->theta, u, h->adam_m,
                       h->adam_v, do_adam, st, loss_out, grad_out, hist_row);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
int launch_dist_finalize(gpimhip_ctx* h, const gpimhip_model_t* m, int64_t N, const double* S, double q2, double lg,
                         double* u, int do_adam, AdamStep st, double* loss_out, double* grad_out, double* hist_row) {
    hipLaunchKernelGGL(dist_finalize_kernel, dim3(1), dim3(64), 0, h->stream, *m, N, S, q2, lg, h->theta, u, h->adam_m,
                       h->adam_v, do_adam, st, loss_out, grad_out, hist_row);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// Symmetry-reduced exact GP on a complete uniform grid (reconstructor(structured=True) with a kernel that does not
// factorise over the axes: role of the reference's structured class gpim/gpreg/skgpr.py:399-448, exact instead of SKI).
// A stationary kernel that is even in every coordinate difference commutes with the reflections g of the grid about
// its centre planes; in the basis  v_{s,p} = |G|^-1/2 sum_g chi_s(g) e_{g p}  (p in the fundamental domain: the first
// half of every reflected axis, s a sign pattern, chi_s(g) = prod_{k in g} s_k) the covariance is block diagonal with
//     K_s[p, q] = sum_g chi_s(g) k(p, g q)              (N / |G| points each, |G| = 2^(reflected axes))
// and (noise + jitter) I stays what it is.  The |G| blocks are the problems of ONE lock-step batch (problem b <-> sign
// pattern: bit j of b = the sign of the j-th reflected axis is -1) with ONE set of hyper-parameters: their gradient sums,
// quadratic forms and log-determinants add up (finalize_coupled_kernel).  Cost of the O(N^3) stages: |G|^-2 of the dense
// model -- 1/16 in two dimensions.  Exact up to rounding: an orthogonal change of basis.
// ------------------------------------------------------------------------------------------
struct ReflPair {
    double dm2[GPIMHIP_MAX_DIM], dp2[GPIMHIP_MAX_DIM];       // squared scaled differences to z and to its mirror image
};
// the dimensions whose sign is -1 in the block of problem pb: bit j of pb belongs to the j-th reflected dimension
__device__ __forceinline__ int refl_sign_dims(int mask, int pb) {
    int sg = 0, j = 0;
#pragma unroll
    for (int k = 0; k < GPIMHIP_MAX_DIM; ++k)
        if ((mask >> k) & 1) {
            if ((pb >> j) & 1) sg |= 1 << k;
            ++j;
        }
    return sg;
}
// f(g, chi, r2) for every reflection g (a subset of mask, as a bit mask over the dimensions); constant trip counts and a
// wave-uniform skip, so that everything stays in registers (unused dimensions hold zeros in p)
template <typename F>
__device__ __forceinline__ void refl_for_each(const ReflPair& p, int mask, int sg, F f) {
#pragma unroll
    for (int g = 0; g < (1 << GPIMHIP_MAX_DIM); ++g) {
        if (g & ~mask) continue;
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) r2 += ((g >> k) & 1) ? p.dp2[k] : p.dm2[k];
        f(g, (__popc(g & sg) & 1) ? -1.0 : 1.0, r2);
    }
}
template <int KIND>
__global__ __launch_bounds__(256) void kmat_refl_kernel(const double* __restrict__ X, int64_t N, const double* __restrict__ Z,
                                                        int64_t M, int d, const ThetaDev* __restrict__ th, double* __restrict__ out,
                                                        int64_t ld, int ntc, int sym, int64_t x_bs, int64_t z_bs, int64_t out_bs,
                                                        ReflArgs refl, double scale) {
    __shared__ double xa[128][4];
    __shared__ double xz[128][4];
    __shared__ double wr_s[128], wc_s[128];
    const int tid = threadIdx.x;
    const int sg = refl_sign_dims(refl.mask, refl.pb_off + (int)blockIdx.y * refl.pb_stride);
    X += blockIdx.y * x_bs;
    Z += blockIdx.y * z_bs;
    th += blockIdx.y;
    out += blockIdx.y * out_bs;
    const double* wts = refl.wts ? refl.wts + (int64_t)blockIdx.y * N : nullptr;
    int ci, cj;
    if (sym) lower_tile_from_linear(blockIdx.x, ci, cj);       // (the factorisation reads the lower tiles only)
    else { ci = blockIdx.x / ntc; cj = blockIdx.x % ntc; }
    const ThetaDev t = *th;
    {
        const bool isrow = tid < 128;
        const int loc = tid & 127;
        const int64_t g = (int64_t)(isrow ? ci : cj) * 128 + loc;
        const double* src = isrow ? X : Z;
        const int64_t lim = isrow ? N : M;
        double (*dst)[4] = isrow ? xa : xz;
#pragma unroll
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) dst[loc][k] = (k < d && g < lim) ? src[g * d + k] / t.ls[k] : 0.0;
        // weights of the block's points (rows; the columns too when they are the same points)
        (isrow ? wr_s : wc_s)[loc] = (wts && (isrow || sym) && g < lim) ? wts[g] : 1.0;
    }
    __syncthreads();
    double cz[GPIMHIP_MAX_DIM];
#pragma unroll
    for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) cz[k] = (k < d) ? refl.twoc[k] / t.ls[k] : 0.0;
    const int ty = tid >> 4, tx = tid & 15;
    for (int rr = 0; rr < 8; ++rr) {
        const int r = ty + 16 * rr;
        const int64_t gi = (int64_t)ci * 128 + r;
        const double a0 = xa[r][0], a1 = xa[r][1], a2 = xa[r][2], a3 = xa[r][3];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            d2 v;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = tx * 2 + 32 * cc + e;
                const int64_t gj = (int64_t)cj * 128 + c;
                const double av[4] = {a0, a1, a2, a3};
                ReflPair p;
#pragma unroll
                for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
                    const double dm = av[k] - xz[c][k], dp = (av[k] + xz[c][k]) - cz[k];
                    p.dm2[k] = dm * dm;
                    p.dp2[k] = dp * dp;
                }
                double acc = 0.0;
                refl_for_each(p, refl.mask, sg, [&](int, double chi, double r2) { acc = fma(chi, kfun_value<KIND>(r2, t.alpha), acc); });
                const double ww = wr_s[r] * wc_s[c];
                double k = scale * t.var * acc * ww;
                if (gi >= N || gj >= M || (sym && ww == 0.0)) k = (sym && gi == gj) ? 1.0 : 0.0;      // padding / absent points
                else if (sym && gi == gj) k += t.diag_add;
                v[e] = k;
            }
            *reinterpret_cast<d2*>(out + gi * ld + (int64_t)cj * 128 + tx * 2 + 32 * cc) = v;
        }
    }
}
int launch_kmat_refl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t N, const double* Z, int64_t M,
                     const ThetaDev* theta, double* out, int64_t ld, int64_t rows_pad, int64_t cols_pad, int sym, int64_t x_bs,
                     int64_t z_bs, int64_t out_bs, double scale) {
    const int ntr = (int)(rows_pad / 128), ntc = (int)(cols_pad / 128);
    if (ntr <= 0 || ntc <= 0) return GPIMHIP_OK;
    const double* Zp = Z ? Z : X;
    if (!Z) z_bs = x_bs;
    dim3 grid((unsigned)(sym ? ntr * (ntr + 1) / 2 : ntr * ntc), h->nbatch), block(256);
#define KR_LAUNCH(KIND)                                                                                                 \
    hipLaunchKernelGGL((kmat_refl_kernel<KIND>), grid, block, 0, h->stream, X, N, Zp, M, m->dim, theta, out, ld, ntc, sym, x_bs, \
                       z_bs, out_bs, h->refl, scale)
    switch (m->kernel) {
        case GPIMHIP_KERNEL_RBF: KR_LAUNCH(GPIMHIP_KERNEL_RBF); break;
        case GPIMHIP_KERNEL_MATERN52: KR_LAUNCH(GPIMHIP_KERNEL_MATERN52); break;
        case GPIMHIP_KERNEL_RQ: KR_LAUNCH(GPIMHIP_KERNEL_RQ); break;
        default: gpim_set_error("unknown kernel kind"); return GPIMHIP_E_BADARG;
    }
#undef KR_LAUNCH
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
// the gradient sums of grad_reduce_kernel for one block K_s: the same contraction with every entry's kernel value and
// derivative factors summed over the reflections (a mirror image's scaled difference (a_k + b_k - c_k / l_k) scales with
// 1 / l_k like the plain one)
template <int KIND>
__global__ __launch_bounds__(256) void grad_reduce_refl_kernel(const double* __restrict__ Kinv, int64_t ld,
                                                               const double* __restrict__ X, int64_t N, int d,
                                                               const double* __restrict__ alpha,
                                                               const ThetaDev* __restrict__ th, double* __restrict__ part,
                                                               int64_t x_bs, int64_t np, ReflArgs refl) {
    __shared__ double xa[128][4];
    __shared__ double xz[128][4];
    __shared__ double al_r[128], al_c[128];
    __shared__ double wr_s[128], wc_s[128];
    __shared__ double red[4][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sg = refl_sign_dims(refl.mask, refl.pb_off + (int)blockIdx.y * refl.pb_stride);
    const double* wts = refl.wts ? refl.wts + (int64_t)blockIdx.y * N : nullptr;
    Kinv += blockIdx.y * np * ld;
    X += blockIdx.y * x_bs;
    alpha += blockIdx.y * np;
    th += blockIdx.y;
    part += (int64_t)blockIdx.y * gridDim.x * 8;
    int ci, cj;
    lower_tile_from_linear(blockIdx.x, ci, cj);
    const ThetaDev t = *th;
    {
        const bool isrow = tid < 128;
        const int loc = tid & 127;
        const int64_t g = (int64_t)(isrow ? ci : cj) * 128 + loc;
        double (*dst)[4] = isrow ? xa : xz;
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) dst[loc][k] = (k < d && g < N) ? X[g * d + k] / t.ls[k] : 0.0;
        (isrow ? al_r : al_c)[loc] = (g < N) ? alpha[g] : 0.0;
        (isrow ? wr_s : wc_s)[loc] = (wts && g < N) ? wts[g] : 1.0;
    }
    __syncthreads();
    double cz[GPIMHIP_MAX_DIM];
    for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) cz[k] = refl.twoc[k] / t.ls[k];
    double S[7] = {0, 0, 0, 0, 0, 0, 0};
    const int ty = tid >> 4, tx = tid & 15;
    for (int rr = 0; rr < 8; ++rr) {
        const int r = ty + 16 * rr;
        const int64_t gi = (int64_t)ci * 128 + r;
        const double ali = al_r[r];
        for (int cc = 0; cc < 8; ++cc) {
            const int c = tx + 16 * cc;
            const int64_t gj = (int64_t)cj * 128 + c;
            if (gi >= N || gj > gi) continue;
            const double g = Kinv[gi * ld + gj] - ali * al_c[c];
            const double ww = wr_s[r] * wc_s[c];
            if (ww == 0.0) continue;                   // (a point that does not exist in this block: an identity row)
            const double w = ((gi == gj) ? g : 2.0 * g) * ww;
            ReflPair p;
            for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
                const double dm = xa[r][k] - xz[c][k], dp = (xa[r][k] + xz[c][k]) - cz[k];
                p.dm2[k] = dm * dm;
                p.dp2[k] = dp * dp;
            }
            refl_for_each(p, refl.mask, sg, [&](int gm, double chi, double r2) {
                const KVal kv = kfun_grad<KIND>(r2, t.alpha);
                const double wc = w * chi;
                S[0] = fma(wc, kv.e, S[0]);
                const double wh = wc * kv.h;
#pragma unroll
                for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) S[1 + k] = fma(wh, ((gm >> k) & 1) ? p.dp2[k] : p.dm2[k], S[1 + k]);
                if (KIND == GPIMHIP_KERNEL_RQ) S[6] = fma(wc, kv.ga, S[6]);
            });
            if (gi == gj) S[5] += g;
        }
    }
    for (int k = 0; k < 7; ++k) {
        const double v = wave_sum(S[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < 8) {
        double v = 0.0;
        if (tid < 7) v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        part[(int64_t)blockIdx.x * 8 + tid] = v;
    }
}
int launch_grad_reduce_refl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld, const double* X,
                            int64_t N, int nb, const double* alpha, int64_t x_bs) {
    const int ntile = nb * (nb + 1) / 2;
    dim3 grid(ntile, h->nbatch), block(256);
#define GR_LAUNCH(KIND)                                                                                              \
    hipLaunchKernelGGL((grad_reduce_refl_kernel<KIND>), grid, block, 0, h->stream, Kinv, ld, X, N, m->dim, alpha, h->theta, \
                       h->grad_part, x_bs, h->np, h->refl)
    switch (m->kernel) {
        case GPIMHIP_KERNEL_RBF: GR_LAUNCH(GPIMHIP_KERNEL_RBF); break;
        case GPIMHIP_KERNEL_MATERN52: GR_LAUNCH(GPIMHIP_KERNEL_MATERN52); break;
        case GPIMHIP_KERNEL_RQ: GR_LAUNCH(GPIMHIP_KERNEL_RQ); break;
        default: gpim_set_error("unknown kernel kind"); return GPIMHIP_E_BADARG;
    }
#undef GR_LAUNCH
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
// finalize for the coupled blocks: the sums of all B problems (fixed order: problem by problem), ONE loss / gradient /
// Adam step on the parameters of problem 0, which are then copied to the other problems' slots
__global__ __launch_bounds__(256) void finalize_coupled_kernel(gpimhip_model_t m, int64_t n_total, int64_t np, int nb, int ntile, int B,
                                                               const double* __restrict__ grad_part, const double* __restrict__ z,
                                                               const double* __restrict__ zb,
                                                               const double* __restrict__ logdet_part,
                                                               const ThetaDev* __restrict__ th, double* __restrict__ u,
                                                               double* __restrict__ adam_m, double* __restrict__ adam_v,
                                                               int do_adam, AdamStep st, double* __restrict__ loss_out,
                                                               double* __restrict__ grad_out, double* __restrict__ hist_row,
                                                               FinalizeIter fi, int32_t* __restrict__ info) {
    __shared__ double red[256];
    __shared__ double S[8];
    const int tid = threadIdx.x;
    const int P = 2 + m.n_ls + (m.kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    double q2 = 0.0, lg = 0.0;
    if (tid < 8) S[tid] = 0.0;
    __syncthreads();
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < 7; ++k) {
            double v = 0.0;
            for (int q = tid; q < ntile; q += 256) v += grad_part[((int64_t)b * ntile + q) * 8 + k];
            v = block_sum_256(v, red);
            if (tid == 0) S[k] += v;
        }
        double qb = 0.0;
        for (int64_t i = tid; i < np; i += 256) qb = fma(z[b * np + i], zb[b * np + i], qb);
        q2 += block_sum_256(qb, red);
        double lb = 0.0;
        for (int k = tid; k < nb; k += 256) lb += logdet_part[(int64_t)b * nb + k];
        lg += block_sum_256(lb, red);
    }
    if (tid != 0) return;
    if (fi.iter) {
        const int it = *fi.iter;
        if (*info != 0) {
            atomicMin(info + 1, it);
            return;
        }
        st.lr_over_bc1 = fi.bc[it];
        st.bc2_sqrt = fi.bc[fi.T + it];
        loss_out = fi.loss_base ? fi.loss_base + it : nullptr;
        hist_row = fi.hist_base ? fi.hist_base + (int64_t)it * P : nullptr;
        *fi.iter = it + 1;
    }
    finalize_step(m, n_total, S, q2, lg, *th, u, adam_m, adam_v, do_adam, st, loss_out, grad_out, hist_row, prior_constant(m));
    for (int b = 1; b < B; ++b)
        for (int k = 0; k < P; ++k) u[(int64_t)b * P + k] = u[k];
}
int launch_finalize_coupled(gpimhip_ctx* h, const gpimhip_model_t* m, int64_t N, int64_t np, double* u, int do_adam,
                            AdamStep st, double* loss_out, double* grad_out, double* hist_row, int32_t* iter,
                            const double* bc, int T, double* hist_base, double* loss_base) {
    const int nb = (int)(np / NB);
    FinalizeIter fi{iter, bc, T, hist_base, loss_base};
    hipLaunchKernelGGL(finalize_coupled_kernel, dim3(1), dim3(256), 0, h->stream, *m, h->refl.n_total > 0 ? h->refl.n_total : N * h->nbatch, np, nb,
                       nb * (nb + 1) / 2, h->nbatch,
                       h->grad_part, h->z, h->fp32 ? h->alpha : h->z, h->logdet_part, h->theta, u, h->adam_m, h->adam_v, do_adam,
                       st, loss_out, grad_out, hist_row, fi, h->info);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
// the sums of the LOCAL blocks, for a job that deals the blocks to its ranks (gpimhip_refl_sums): out[0..7] = the gradient
// sums, out[8] = sum log L_ii, out[9] = 1 if a factorisation failed, out[10] = sum |L^-1 y|^2 -- fixed order, problem by problem
__global__ __launch_bounds__(256) void coupled_sums_kernel(int64_t np, int nb, int ntile, int B, const double* __restrict__ grad_part,
                                                           const double* __restrict__ z, const double* __restrict__ logdet_part,
                                                           const int32_t* __restrict__ info, double* __restrict__ out) {
    __shared__ double red[256];
    __shared__ double S[8];
    const int tid = threadIdx.x;
    double q2 = 0.0, lg = 0.0;
    if (tid < 8) S[tid] = 0.0;
    __syncthreads();
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < 7; ++k) {
            double v = 0.0;
            for (int q = tid; q < ntile; q += 256) v += grad_part[((int64_t)b * ntile + q) * 8 + k];
            v = block_sum_256(v, red);
            if (tid == 0) S[k] += v;
        }
        double qb = 0.0;
        for (int64_t i = tid; i < np; i += 256) qb = fma(z[b * np + i], z[b * np + i], qb);
        q2 += block_sum_256(qb, red);
        double lb = 0.0;
        for (int k = tid; k < nb; k += 256) lb += logdet_part[(int64_t)b * nb + k];
        lg += block_sum_256(lb, red);
    }
    if (tid != 0) return;
    for (int k = 0; k < 8; ++k) out[k] = S[k];
    out[8] = lg;
    out[9] = (*info != 0) ? 1.0 : 0.0;
    out[10] = q2;
}
int launch_coupled_sums(gpimhip_ctx* h, int64_t np, double* out11) {
    const int nb = (int)(np / NB);
    hipLaunchKernelGGL(coupled_sums_kernel, dim3(1), dim3(256), 0, h->stream, np, nb, nb * (nb + 1) / 2, h->nbatch, h->grad_part,
                       h->z, h->logdet_part, h->info, out11);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
// posterior of the coupled blocks: mean_j = sum_b mean_b[j], var_j = clamp(s2 - sum_b sum_ci colpart_b[ci][j], 0) + noise
// (the variance only for the chunk's first nvar test points: ReflArgs::var_count)
__global__ void predict_coupled_kernel(const double* __restrict__ colpart, int64_t ldp, int nb, int B, int64_t m0, int64_t mcount,
                                       int64_t nvar, const double* __restrict__ mean_tmp, int64_t mean_bs,
                                       const ThetaDev* __restrict__ th, double* __restrict__ mean_out, double* __restrict__ var_out,
                                       int raw) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= mcount) return;
    double q = 0.0, mu = 0.0;
    for (int b = 0; b < B; ++b) {
        if (j < nvar)
            for (int ci = 0; ci < nb; ++ci) q += colpart[((int64_t)b * nb + ci) * ldp + j];
        mu += mean_tmp[b * mean_bs + j];
    }
    mean_out[m0 + j] = mu;
    if (j < nvar) var_out[m0 + j] = raw ? q : clamp0_nan(th->var - q) + th->noise;
}
int launch_predict_coupled(gpimhip_ctx* h, int64_t ldp, int nb, int64_t m0, int64_t mcount, int64_t nvar, int64_t mean_bs,
                           double* mean_out, double* var_out) {
    hipLaunchKernelGGL(predict_coupled_kernel, dim3((unsigned)((mcount + 255) / 256)), dim3(256), 0, h->stream, h->colpart, ldp,
                       nb, h->nbatch, m0, mcount, nvar, h->mean_tmp, mean_bs, h->theta, mean_out, var_out, h->refl.raw);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// The diagonal of a column slab of the covariance (columns row0 .. row0 + npad - 1 of the padded matrix):
// out[(row0 + j) * ld + j] += jitter + noise(theta) for the n valid columns, = 1 for the padding columns
__global__ void add_diag_theta_kernel(double* __restrict__ out, int64_t ld, int64_t row0, int64_t n, int64_t npad,
                                      const ThetaDev* __restrict__ th) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[(row0 + j) * ld + j] += th->diag_add;
    else if (j < npad) out[(row0 + j) * ld + j] = 1.0;
}
int launch_add_diag_theta(gpimhip_ctx* h, double* out, int64_t ld, int64_t row0, int64_t n, int64_t npad) {
    hipLaunchKernelGGL(add_diag_theta_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, h->stream, out, ld, row0, n,
                       npad, h->theta);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
