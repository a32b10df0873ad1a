// potf2.hip -- one workgroup factors a 128x128 diagonal block entirely in LDS and also
// produces its explicit inverse.
//
// Role in the hot path: the diagonal step of the blocked right-looking
// Cholesky that replaces torch.linalg.cholesky (call sites gpim/gpreg/gpr.py:192,248).  The
// inverse turns the panel triangular solves and the leaves of the triangular inversion into
// MFMA GEMMs.
//
// Inside the workgroup the block is processed in 16-column panels:
//   wave 0 / 16 lanes: 16x16 Cholesky + inverse held in registers (row per lane, cross-lane
//                      broadcasts by v_readlane -- no LDS round trips on the serial chain)
//   all waves:         panel solve  P <- P * inv(L16)^T         (v_mfma_f64_16x16x4_f64)
//   all waves:         trailing     D <- D - P P^T  (lower)     (v_mfma_f64_16x16x4_f64)
// then inv(L) by recursive doubling (16 -> 32 -> 64 -> 128), products on MFMA, done in place.
#include "common.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));

#ifdef POTF2_PROFILE
#define PROF_ARG , long long* __restrict__ prof
#define STAMP(i) do { if (threadIdx.x == 0) prof[i] = clock64(); } while (0)
#else
#define PROF_ARG
#define STAMP(i) do { } while (0)
#endif

#define LDD 130   // row stride of the LDS block: 130 % 32 == 2 keeps MK fragment reads conflict free

__device__ __forceinline__ double bcast_lane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// 16x16 lower Cholesky + inverse by lanes 0..15 of one wave.  D points at the (c0,c0) corner.
// On exit D holds L16 (lower), I16 (stride 17) holds inv(L16) with explicit zeros above the diagonal.
// Returns 0, or 1 + local column of the first non-positive pivot.
__device__ __forceinline__ int chol16_inv(double* D, double* I16, int lane) {
    const int r = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = D[r * LDD + c];
    double invd[16];
    int bad = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double djj = bcast_lane(a[j], j);
        if (!(djj > 0.0) && bad == 0) bad = j + 1;
        // 1/sqrt(d) by v_rsq_f64 + one Newton step each for l = sqrt(d) and 1/l: a third of the
        // dependent-instruction chain of sqrt() followed by a division, same accuracy class
        double inv = rsqrt(djj);
        double ljj = djj * inv;
        ljj = fma(0.5 * inv, fma(-ljj, ljj, djj), ljj);
        inv = fma(inv, fma(-ljj, inv, 1.0), inv);
        invd[j] = inv;
        a[j] = (r == j) ? ljj : a[j] * inv;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
            const double lcj = bcast_lane(a[j], c);
            a[c] -= a[j] * lcj;
        }
    }
    // inverse: lane c builds column c of X = inv(L): X[r][c] = (delta_rc - sum_{k<r} L[r][k] X[k][c]) / L[r][r]
    double xc[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        double s = (rr == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) {
            const double lrk = bcast_lane(a[k], rr);
            s -= lrk * xc[k];
        }
        xc[rr] = s * invd[rr];
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) D[r * LDD + c] = (c <= r) ? a[c] : 0.0;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) I16[rr * 17 + r] = (rr >= r) ? xc[rr] : 0.0;
    }
    return bad;
}

// C(16x16 at D[r0.., c0..]) (op)= sum over kdepth of A(rows ar0.., k from ak0) * B
// A is read "MK" from D; B either "MK" (NT: B[n][k] at D[(bn0+n)*ldb + bk0+k]) or "KM".
__device__ __forceinline__ d4 tile_mma(d4 acc, const double* Ab, int lda, const double* Bb, int ldb,
                                       bool b_km, int kdepth, int lane, double asign) {
    for (int s = 0; s < kdepth; s += 4) {
        const double a = asign * Ab[(lane & 15) * lda + s + (lane >> 4)];
        const double b = b_km ? Bb[(s + (lane >> 4)) * ldb + (lane & 15)]
                              : Bb[(lane & 15) * ldb + s + (lane >> 4)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ d4 tile_read(const double* C, int lane) {
    d4 v;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) v[rg] = C[((lane >> 4) + 4 * rg) * LDD + (lane & 15)];
    return v;
}
__device__ __forceinline__ void tile_write(double* C, d4 v, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) C[((lane >> 4) + 4 * rg) * LDD + (lane & 15)] = v[rg];
}

// A: matrix (row-major, ld); kblk: which diagonal block.  dinv_out: 128x128 (ld 128) inverse.
// logdet_out[kblk] = sum_i log L_ii over the block.  info: first failing column + 1 (set once).
__global__ __launch_bounds__(256, 1) void potf2_inv_kernel(double* __restrict__ A, int64_t ld, int kblk,
                                                           double* __restrict__ dinv_all,
                                                           double* __restrict__ logdet_out,
                                                           int32_t* __restrict__ info PROF_ARG) {
    __shared__ __attribute__((aligned(16))) double D[NB * LDD];
    __shared__ __attribute__((aligned(16))) double I16[8 * 16 * 17];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double* Ablk = A + ((int64_t)kblk * NB) * ld + (int64_t)kblk * NB;
    if (tid == 0) s_bad = 0;
    STAMP(0);
    for (int e = tid; e < NB * (NB / 2); e += 256) {
        const int r = e >> 6, c2 = (e & 63) * 2;
        const double2 v = *reinterpret_cast<const double2*>(Ablk + (int64_t)r * ld + c2);
        D[r * LDD + c2] = v.x;
        D[r * LDD + c2 + 1] = v.y;
    }
    __syncthreads();
    STAMP(1);

    for (int p = 0; p < 8; ++p) {
        const int c0 = p * 16;
        STAMP(2 + 3 * p);
        if (wave == 0) {
            int bad = chol16_inv(D + c0 * LDD + c0, I16 + p * 16 * 17, lane);
            if (lane == 0 && bad && s_bad == 0) s_bad = c0 + bad;
        }
        __syncthreads();
        STAMP(3 + 3 * p);
        // panel solve: rows below the diagonal tile
        for (int rt = p + 1 + wave; rt < 8; rt += 4) {
            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
            acc = tile_mma(acc, D + rt * 16 * LDD + c0, LDD, I16 + p * 16 * 17, 17, false, 16, lane, 1.0);
            tile_write(D + rt * 16 * LDD + c0, acc, lane);
        }
        __syncthreads();
        STAMP(4 + 3 * p);
        // trailing update of the lower tiles (rt >= ct > p); each wave keeps up to four independent
        // accumulators in flight (one dependent fp64 MFMA chain alone leaves the pipe mostly idle)
        const int m = 7 - p;                 // tiles per side
        const int ntile = m * (m + 1) / 2;
        for (int base = 0; base < ntile; base += 16) {
            d4 acc[4];
            int rt[4], ct[4];
            bool on[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = base + wave + 4 * u;
                on[u] = q < ntile;
                int i = 0, j = 0;
                if (on[u]) {
                    i = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
                    while (i * (i + 1) / 2 > q) --i;
                    while ((i + 1) * (i + 2) / 2 <= q) ++i;
                    j = q - i * (i + 1) / 2;
                }
                rt[u] = p + 1 + i;
                ct[u] = p + 1 + j;
                acc[u] = on[u] ? tile_read(D + rt[u] * 16 * LDD + ct[u] * 16, lane) : (d4){0.0, 0.0, 0.0, 0.0};
            }
#pragma unroll
            for (int sft = 0; sft < 16; sft += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (!on[u]) continue;
                    const double a = -D[(rt[u] * 16 + (lane & 15)) * LDD + c0 + sft + (lane >> 4)];
                    const double b = D[(ct[u] * 16 + (lane & 15)) * LDD + c0 + sft + (lane >> 4)];
                    acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (on[u]) tile_write(D + rt[u] * 16 * LDD + ct[u] * 16, acc[u], lane);
        }
        __syncthreads();
    }

    STAMP(26);
    // write L (zeros above the diagonal) back, and the log-determinant partial
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 7, c = e & 127;
        Ablk[(int64_t)r * ld + c] = (c <= r) ? D[r * LDD + c] : 0.0;
    }
    double lg = 0.0;
    if (tid < 128) lg = log(D[tid * LDD + tid]);
    __syncthreads();
    // --- inverse by recursive doubling, in place in D ---
    // level 0: diagonal 16x16 blocks <- inv(L16), zeros above
    for (int e = tid; e < 8 * 256; e += 256) {
        const int p = e >> 8, rr = (e >> 4) & 15, c = e & 15;
        D[(p * 16 + rr) * LDD + p * 16 + c] = I16[p * 16 * 17 + rr * 17 + c];
    }
    __syncthreads();
    // log-det reduction through LDS (I16 is free now)
    double* red = I16;
    if (tid < 128) red[tid] = lg;
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
    for (int half = 16; half <= 64; half *= 2) {
        const int ht = half / 16;               // tiles per side of one sub-block
        const int tiles_per_pair = ht * ht;
        const int ntile = (NB / (2 * half)) * tiles_per_pair;   // 4, 8, 16  (<= 4 per wave)
        d4 keep[4];
        // phase 1: T = L21 * X11  (X11 lower: k-tiles kt >= tj); results stay in registers until
        // every wave has finished reading L21, then overwrite it.
        int ti4[4], tj4[4], r04[4], c04[4];
        bool on4[4];
#pragma unroll
        for (int cnt = 0; cnt < 4; ++cnt) {
            const int q = wave + 4 * cnt;
            on4[cnt] = q < ntile;
            const int pr = q / tiles_per_pair, w = q % tiles_per_pair;
            ti4[cnt] = w / ht;
            tj4[cnt] = w % ht;
            r04[cnt] = (2 * pr + 1) * half;
            c04[cnt] = 2 * pr * half;
            keep[cnt] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        for (int kt = 0; kt < ht; ++kt)
#pragma unroll
            for (int sft = 0; sft < 16; sft += 4)
#pragma unroll
                for (int cnt = 0; cnt < 4; ++cnt) {
                    if (!on4[cnt] || kt < tj4[cnt]) continue;
                    const double a = D[(r04[cnt] + ti4[cnt] * 16 + (lane & 15)) * LDD + c04[cnt] + kt * 16 + sft + (lane >> 4)];
                    const double b = D[(c04[cnt] + kt * 16 + sft + (lane >> 4)) * LDD + c04[cnt] + tj4[cnt] * 16 + (lane & 15)];
                    keep[cnt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, keep[cnt], 0, 0, 0);
                }
        __syncthreads();
#pragma unroll
        for (int cnt = 0; cnt < 4; ++cnt) {
            const int q = wave + 4 * cnt;
            if (q < ntile) {
                const int pr = q / tiles_per_pair, w = q % tiles_per_pair;
                const int ti = w / ht, tj = w % ht;
                const int r0 = (2 * pr + 1) * half, c0 = 2 * pr * half;
                tile_write(D + (r0 + ti * 16) * LDD + c0 + tj * 16, keep[cnt], lane);
            }
        }
        __syncthreads();
        // phase 2: X21 = -X22 * T  (X22 lower: k-tiles kt <= ti)
#pragma unroll
        for (int cnt = 0; cnt < 4; ++cnt) keep[cnt] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int kt = 0; kt < ht; ++kt)
#pragma unroll
            for (int sft = 0; sft < 16; sft += 4)
#pragma unroll
                for (int cnt = 0; cnt < 4; ++cnt) {
                    if (!on4[cnt] || kt > ti4[cnt]) continue;
                    const double a = -D[(r04[cnt] + ti4[cnt] * 16 + (lane & 15)) * LDD + r04[cnt] + kt * 16 + sft + (lane >> 4)];
                    const double b = D[(r04[cnt] + kt * 16 + sft + (lane >> 4)) * LDD + c04[cnt] + tj4[cnt] * 16 + (lane & 15)];
                    keep[cnt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, keep[cnt], 0, 0, 0);
                }
        __syncthreads();
#pragma unroll
        for (int cnt = 0; cnt < 4; ++cnt) {
            const int q = wave + 4 * cnt;
            if (q < ntile) {
                const int pr = q / tiles_per_pair, w = q % tiles_per_pair;
                const int ti = w / ht, tj = w % ht;
                const int r0 = (2 * pr + 1) * half, c0 = 2 * pr * half;
                tile_write(D + (r0 + ti * 16) * LDD + c0 + tj * 16, keep[cnt], lane);
            }
        }
        __syncthreads();
    }
    STAMP(28);
    double* dinv = dinv_all + (int64_t)kblk * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e >> 7, c = e & 127;
        dinv[e] = (c <= r) ? D[r * LDD + c] : 0.0;
    }
    STAMP(29);
}

#ifndef POTF2_PROFILE
int launch_potf2(gpimhip_ctx* h, double* A, int64_t ld, int kblk, int32_t* info) {
    hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), 0, h->stream, A, ld, kblk, h->dinv,
                       h->logdet_part, info);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
#endif
