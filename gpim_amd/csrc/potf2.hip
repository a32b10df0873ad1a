// potf2.hip -- the diagonal-block kernels of the blocked Cholesky (replaces
// the diagonal steps of torch.linalg.cholesky, call sites gpim/gpreg/gpr.py:192,248).
//
//   potf2_kernel   one workgroup (8 waves) factors a 128x128 diagonal block that is resident in
//                  LDS and then inverts the factor in place (recursive doubling 16 -> 32 -> 64 -> 128
//                  on MFMA).  The explicit inverse turns the panel triangular solve and the leaves of
//                  the triangular inversion into GEMMs.
//
// potf2 works in 16-column panels:
//   wave 0, 16 lanes : 16x16 Cholesky in registers (row per lane; cross-lane broadcasts by
//                      v_readlane, pivots by v_rsq_f64 + Newton) -- the only serial chain
//   threads, 1/row   : panel solve by forward substitution against the 16x16 factor (LDS broadcast)
//   8 waves          : trailing update D -= P P^T on the lower 16x16 tiles, v_mfma_f64_16x16x4_f64
// A lone wave can issue an fp64 MFMA only every ~140 cycles, so the MFMA phases need >= 2 waves per
// SIMD: hence 512-thread workgroups.
#include "blocklds.hpp"

#ifdef POTF2_PROFILE
#define PROF_ARG , long long* __restrict__ prof
#define STAMP(i) do { if (threadIdx.x == 0) prof[i] = clock64(); } while (0)
#else
#define PROF_ARG
#define STAMP(i) do { } while (0)
#endif


// A: matrix (row-major, ld); kblk: which diagonal block; blockIdx.y: problem of a batch.
// dinv_all[kblk] <- inverse of the factored block (ld 128, zeros above the diagonal).
// logdet_out[kblk] = sum_i log L_ii.  info: 1 + first failing global column (set once).
template <typename R>
__global__ __launch_bounds__(NTH, 1) void potf2_kernel(R* __restrict__ A, int64_t ld, int kblk,
                                                       R* __restrict__ dinv_all,
                                                       double* __restrict__ logdet_out,
                                                       int32_t* __restrict__ info, int nb PROF_ARG) {
    __shared__ __attribute__((aligned(16))) double D[NB * LDD];
    __shared__ double invd[NB];
    __shared__ double Xs[16 * XS_LD];
    __shared__ double red[2];
    __shared__ int s_bad;
    A += (int64_t)blockIdx.y * nb * NB * ld;
    dinv_all += (int64_t)blockIdx.y * nb * NB * NB;
    logdet_out += (int64_t)blockIdx.y * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    typedef R RV2 __attribute__((ext_vector_type(2)));
    R* Ablk = A + ((int64_t)kblk * NB) * ld + (int64_t)kblk * NB;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    STAMP(0);
    load_block_chol0(D, invd, &s_bad, Ablk, ld, tid);
    STAMP(1);
    // factor and invert in one sweep; block row i of L goes back to HBM (zeros above the diagonal)
    // during step i, just before the inverse overwrites it in LDS
    auto export_row = [&](int i, int t) {
        for (int e = t; e < 16 * 64; e += SINK_THREADS) {
            const int r = i * 16 + (e >> 6), c = (e & 63) * 2;
            RV2 v;
            v[0] = (R)((c <= r) ? D[r * LDD + c] : 0.0);
            v[1] = (R)((c + 1 <= r) ? D[r * LDD + c + 1] : 0.0);
            *reinterpret_cast<RV2*>(Ablk + (int64_t)r * ld + c) = v;
        }
    };
    lds_factor_inv<decltype(export_row), true>(D, invd, Xs, 8, &s_bad, tid, export_row);
    STAMP(2);
    // log-determinant partial (fixed order) from the reciprocal pivots
    if (wave < 2) {
        const double v = wave_sum(-log(invd[tid]));
        if (lane == 0) red[wave] = v;
    }
    __syncthreads();
    if (tid == 0) {
        logdet_out[kblk] = red[0] + red[1];
        if (s_bad != 0 && *info == 0) *info = kblk * NB + s_bad;
    }
    STAMP(3);
    STAMP(4);
    R* dinv = dinv_all + (int64_t)kblk * NB * NB;
    for (int e = tid; e < NB * NB / 2; e += NTH) {
        const int r = e >> 6, c = (e & 63) * 2;
        RV2 v;
        v[0] = (R)((c <= r) ? D[r * LDD + c] : 0.0);
        v[1] = (R)((c + 1 <= r) ? D[r * LDD + c + 1] : 0.0);
        *reinterpret_cast<RV2*>(dinv + r * NB + c) = v;
    }
    STAMP(5);
}

#ifndef POTF2_PROFILE
int launch_potf2(gpimhip_ctx* h, double* A, int64_t ld, int kblk, int32_t* info) {
    const int nb = (int)(h->np / NB);
    if (h->fp32)
        hipLaunchKernelGGL(potf2_kernel<float>, dim3(1, h->nbatch), dim3(NTH), 0, h->stream, reinterpret_cast<float*>(A), ld,
                           kblk, reinterpret_cast<float*>(h->dinv), h->logdet_part, info, nb);
    else
        hipLaunchKernelGGL(potf2_kernel<double>, dim3(1, h->nbatch), dim3(NTH), 0, h->stream, A, ld, kblk, h->dinv,
                           h->logdet_part, info, nb);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
#endif
