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

// ------------------------------------------------------------------------------------------
// O(N^2) vector solves of the distributed model (alpha = K^-1 y: DistributedCholesky.solve), panel by panel on the
// owner of the panel.  The panel's 512 x 512 diagonal triangle is never inverted as a whole: its four 128 x 128
// diagonal blocks come with their explicit inverses (dinv, written by the factorisation role), so the triangular solve
// is a block substitution of 128 x 128 mat-vecs inside ONE workgroup.
//   forward :  piece = Lpp^-1 (y_p - t),   then  acc[rows below] += L(below, p) piece        (trsv + rows kernel)
//   backward:  rhs = z_p - L(below, p)^T a[below]  (cols kernel),  then  piece = Lpp^-T rhs   (trsv)
// ------------------------------------------------------------------------------------------
// block b of the result: x_b = D_b (r_b - sum_{c < b} L_bc x_c)  (forward)  /  x_b = D_b^T (r_b - sum_{c > b} L_cb^T x_c)
// P: the panel's diagonal triangle (w x w, ld), D: dinv blocks of the panel (128 x 128 each, contiguous), 512 threads:
// thread = (row r of the block, quarter q of the 128-long dot product); fixed summation order.
__global__ __launch_bounds__(512) void dist_trsv_kernel(const double* __restrict__ P, int64_t ld, const double* __restrict__ D,
                                                        int nblk, int backward, const double* __restrict__ r0,
                                                        const double* __restrict__ r1, double* __restrict__ out) {
    __shared__ double x[4 * NB];
    __shared__ double v[NB];
    __shared__ double part[4][NB];
    const int tid = threadIdx.x, r = tid & 127, q = tid >> 7;
    for (int e = tid; e < nblk * NB; e += 512) x[e] = 0.0;
    // a ragged last panel (nblk < 4): the entries of `out` (512 doubles, broadcast and copied as a whole by the driver)
    // it does not own are zero, not what the previous panel left there
    for (int e = nblk * NB + tid; e < 4 * NB; e += 512) out[e] = 0.0;
    __syncthreads();
    for (int s = 0; s < nblk; ++s) {
        const int b = backward ? nblk - 1 - s : s;
        // v = r_b - sum_c L x_c
        double acc = 0.0;
        if (!backward) {
            for (int c = 0; c < b; ++c)
                for (int k = q * 32; k < q * 32 + 32; ++k) acc = fma(P[(int64_t)(b * NB + r) * ld + c * NB + k], x[c * NB + k], acc);
        } else {
            for (int c = b + 1; c < nblk; ++c)
                for (int k = q * 32; k < q * 32 + 32; ++k) acc = fma(P[(int64_t)(c * NB + k) * ld + b * NB + r], x[c * NB + k], acc);
        }
        part[q][r] = acc;
        __syncthreads();
        if (q == 0) v[r] = (r0[b * NB + r] - (r1 ? r1[b * NB + r] : 0.0)) - ((part[0][r] + part[1][r]) + (part[2][r] + part[3][r]));
        __syncthreads();
        // x_b = D_b v  (or D_b^T v)
        const double* Db = D + (int64_t)b * NB * NB;
        acc = 0.0;
        for (int k = q * 32; k < q * 32 + 32; ++k) acc = fma(backward ? Db[k * NB + r] : Db[r * NB + k], v[k], acc);
        part[q][r] = acc;
        __syncthreads();
        if (q == 0) {
            const double xv = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
            x[b * NB + r] = xv;
            out[b * NB + r] = xv;
        }
        __syncthreads();
    }
}
// acc[i] += sum_j A[i][j] x[j], j < w (<= 512); one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void dist_rows_acc_kernel(const double* __restrict__ A, int64_t ld, int64_t rows, int w,
                                                            const double* __restrict__ x, double* __restrict__ acc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= rows) return;
    double s = 0.0;
    for (int j = lane * 2; j < w; j += 128) {
        const double2 a = *reinterpret_cast<const double2*>(A + i * ld + j);
        s = fma(a.x, x[j], s);
        s = fma(a.y, x[j + 1], s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) acc[i] += s;
}
int launch_dist_trsv(gpimhip_ctx* h, const double* P, int64_t ld, const double* D, int nblk, int backward, const double* r0,
                     const double* r1, double* out) {
    hipLaunchKernelGGL(dist_trsv_kernel, dim3(1), dim3(512), 0, h->stream, P, ld, D, nblk, backward, r0, r1, out);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
int launch_dist_rows_acc(gpimhip_ctx* h, const double* A, int64_t ld, int64_t rows, int w, const double* x, double* acc) {
    if (rows <= 0) return GPIMHIP_OK;
    hipLaunchKernelGGL(dist_rows_acc_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, h->stream, A, ld, rows, w, x, acc);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
