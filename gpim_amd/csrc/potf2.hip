// potf2.hip -- the diagonal-block kernels of the blocked Cholesky (replaces
// the diagonal steps of torch.linalg.cholesky, call sites gpim/gpreg/gpr.py:192,248).
//
//   potf2_kernel   one workgroup (8 waves) factors a 128x128 diagonal block that is resident in
//                  LDS and also emits the inverses of its eight 16x16 diagonal sub-blocks.
//   inv128_kernel  one workgroup per block: explicit inverse of a factored 128x128 block by
//                  recursive doubling (16 -> 32 -> 64 -> 128) on MFMA.  The inverse turns the panel
//                  triangular solve and the leaves of the triangular inversion into GEMMs.
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


// A: matrix (row-major, ld); kblk: which diagonal block.
// linv16_all[kblk][8][16][16]: inverses of the 16x16 diagonal sub-blocks (zeros above the diagonal).
// logdet_out[kblk] = sum_i log L_ii.  info: 1 + first failing global column (set once).
__global__ __launch_bounds__(NTH, 1) void potf2_kernel(double* __restrict__ A, int64_t ld, int kblk,
                                                       double* __restrict__ linv16_all,
                                                       double* __restrict__ logdet_out,
                                                       int32_t* __restrict__ info, int nb PROF_ARG) {
    __shared__ __attribute__((aligned(16))) double D[NB * LDD];
    __shared__ double invd[NB];
    A += (int64_t)blockIdx.y * nb * NB * ld;
    linv16_all += (int64_t)blockIdx.y * nb * 2048;
    logdet_out += (int64_t)blockIdx.y * nb;
    __shared__ double red[128];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* Ablk = A + ((int64_t)kblk * NB) * ld + (int64_t)kblk * NB;
    if (tid == 0) s_bad = 0;
    STAMP(0);
    load_block(D, Ablk, ld, tid);
    __syncthreads();
    STAMP(1);

    lds_factor(D, invd, 8, &s_bad, tid);
    STAMP(26);
    // L back to HBM (zeros above the diagonal)
    for (int e = tid; e < NB * NB / 2; e += NTH) {
        const int r = e >> 6, c = (e & 63) * 2;
        d2 v;
        v[0] = (c <= r) ? D[r * LDD + c] : 0.0;
        v[1] = (c + 1 <= r) ? D[r * LDD + c + 1] : 0.0;
        *reinterpret_cast<d2*>(Ablk + (int64_t)r * ld + c) = v;
    }
    // inverses of the eight 16x16 diagonal sub-blocks, one wave each
    trinv16(D + wave * 16 * LDD + wave * 16, LDD, invd + wave * 16,
            linv16_all + ((int64_t)kblk * 8 + wave) * 256, 16, lane);
    // log-determinant partial (fixed tree)
    if (tid < 128) red[tid] = log(D[tid * LDD + tid]);
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        logdet_out[kblk] = red[0];
        if (s_bad != 0 && *info == 0) *info = kblk * NB + s_bad;
    }
    STAMP(27);
}

// dinv_all[blk] = inverse of the factored diagonal block blk = kblk0 + blockIdx.x (ld 128, zeros above
// the diagonal), from L (in A) and the 16x16 diagonal inverses.
__global__ __launch_bounds__(NTH, 1) void inv128_kernel(const double* __restrict__ A, int64_t ld, int kblk0,
                                                        const double* __restrict__ linv16_all,
                                                        double* __restrict__ dinv_all, int nb) {
    __shared__ __attribute__((aligned(16))) double D[NB * LDD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kblk = kblk0 + blockIdx.x;
    A += (int64_t)blockIdx.y * nb * NB * ld;
    linv16_all += (int64_t)blockIdx.y * nb * 2048;
    dinv_all += (int64_t)blockIdx.y * nb * NB * NB;
    const double* Ablk = A + ((int64_t)kblk * NB) * ld + (int64_t)kblk * NB;
    load_block(D, Ablk, ld, tid);
    __syncthreads();
    // level 0: diagonal 16x16 blocks <- their inverses
    for (int e = tid; e < 8 * 256; e += NTH) {
        const int p = e >> 8, rr = (e >> 4) & 15, c = e & 15;
        D[(p * 16 + rr) * LDD + p * 16 + c] = linv16_all[((int64_t)kblk * 8 + p) * 256 + rr * 16 + c];
    }
    __syncthreads();
    lds_invert_levels(D, 8, tid);
    double* dinv = dinv_all + (int64_t)kblk * NB * NB;
    for (int e = tid; e < NB * NB / 2; e += NTH) {
        const int r = e >> 6, c = (e & 63) * 2;
        d2 v;
        v[0] = (c <= r) ? D[r * LDD + c] : 0.0;
        v[1] = (c + 1 <= r) ? D[r * LDD + c + 1] : 0.0;
        *reinterpret_cast<d2*>(dinv + r * NB + c) = v;
    }
}

#ifndef POTF2_PROFILE
int launch_potf2(gpimhip_ctx* h, double* A, int64_t ld, int kblk, int32_t* info) {
    const int nb = (int)(h->np / NB);
    hipLaunchKernelGGL(potf2_kernel, dim3(1, h->nbatch), dim3(NTH), 0, h->stream, A, ld, kblk, h->linv16,
                       h->logdet_part, info, nb);
    hipLaunchKernelGGL(inv128_kernel, dim3(1, h->nbatch), dim3(NTH), 0, h->stream, A, ld, kblk, h->linv16, h->dinv, nb);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
#endif
