// distops.hip -- small device kernels of the distributed (block-column-cyclic) exact GP (api.hip: gpimhip_dist_*).
#include "common.hpp"

// Broadcast buffer of one factored panel: rows [r0, np) x w columns of L, and -- in the 128 rows after np -- the
// inverses of the panel's diagonal blocks side by side (128 x 128 each), which the forward substitutions of the
// other ranks need (gpimhip_dist_solve_update).
__global__ __launch_bounds__(256) void dist_pack_kernel(const double* __restrict__ P, int64_t ldp, int64_t r0, int64_t np,
                                                        int w, const double* __restrict__ dinv, int nblk,
                                                        double* __restrict__ buf, int64_t ldb) {
    const int64_t row = r0 + blockIdx.x;            // rows r0 .. np + 127
    if (row < np) {
        for (int c = threadIdx.x * 2; c < w; c += 512)
            *reinterpret_cast<double2*>(buf + row * ldb + c) = *reinterpret_cast<const double2*>(P + row * ldp + c);
    } else {
        const int r = (int)(row - np);
        for (int c = threadIdx.x * 2; c < nblk * NB; c += 512)
            *reinterpret_cast<double2*>(buf + row * ldb + c) =
                *reinterpret_cast<const double2*>(dinv + (int64_t)(c / NB) * NB * NB + (int64_t)r * NB + (c % NB));
    }
}
int launch_dist_pack(gpimhip_ctx* h, const double* P, int64_t ldp, int64_t r0, int64_t np, int w, const double* dinv,
                     int nblk, double* buf, int64_t ldb) {
    hipLaunchKernelGGL(dist_pack_kernel, dim3((unsigned)(np + NB - r0)), dim3(256), 0, h->stream, P, ldp, r0, np, w, dinv,
                       nblk, buf, ldb);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// q[j] += sum_r W[r][j]^2 over `rows` rows (fixed order: one thread per column walks the rows)
__global__ __launch_bounds__(256) void colsumsq_acc_kernel(const double* __restrict__ W, int64_t ldw, int rows, int64_t m,
                                                           double* __restrict__ q) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int r = 0; r < rows; ++r) {
        const double v = W[(int64_t)r * ldw + j];
        s = fma(v, v, s);
    }
    q[j] += s;
}
int launch_colsumsq_acc(gpimhip_ctx* h, const double* W, int64_t ldw, int rows, int64_t m, double* q) {
    hipLaunchKernelGGL(colsumsq_acc_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, W, ldw, rows, m, q);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
