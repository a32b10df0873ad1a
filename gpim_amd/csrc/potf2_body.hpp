// potf2_body.hpp -- the diagonal-block factorisation of the blocked Cholesky as a device function (one 512-thread
// workgroup; see potf2.hip).  Hosted by potf2_kernel and by the step kernel of cholstep.hip.
#pragma once
#include "blocklds.hpp"

#ifdef POTF2_PROFILE
#define PROF_ARG , long long* __restrict__ prof
#define STAMP(i) do { if (threadIdx.x == 0) prof[i] = clock64(); } while (0)
#else
#define PROF_ARG
#define STAMP(i) do { } while (0)
#endif

// A: matrix (row-major, ld); kblk: which diagonal block; by: problem of a batch.
// dinv_all[kblk] <- inverse of the factored block (ld 128, zeros above the diagonal).
// logdet_out[kblk] = sum_i log L_ii.  info: 1 + first failing global column (set once).
// dinvB_all (optional): the same inverse once more in the B-operand order of the f64 MFMA for the product
// S = P * Dinv^T of the panel solve (cholstep.hip): dinvB[t][s2][lane][e] = Dinv[16 t + (lane & 15)][8 s2 + 4 e + (lane >> 4)],
// so that a wave fetches the fragments of two k-steps of one 16-column tile with ONE coalesced 16-byte load per lane.
// smem: POTF2_SMEM_DOUBLES doubles of LDS.
#define POTF2_SMEM_DOUBLES (NB * LDD + NB + 32 * XS_LD + 4)
template <typename R>
__device__ __forceinline__ void potf2_body(double* __restrict__ smem, const int by, R* __restrict__ A, int64_t ld, int kblk,
                                           R* __restrict__ dinv_all, double* __restrict__ dinvB_all,
                                           double* __restrict__ logdet_out, int32_t* __restrict__ info, int nb PROF_ARG,
                                           int col_off = 0) {
    double* D = smem;
    double* invd = D + NB * LDD;
    double* Xs = invd + NB;
    double* red = Xs + 32 * XS_LD;
    int& s_bad = *reinterpret_cast<int*>(red + 2);
    A += (int64_t)by * nb * NB * ld;
    dinv_all += (int64_t)by * nb * NB * NB;
    logdet_out += (int64_t)by * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    typedef R RV2 __attribute__((ext_vector_type(2)));
    R* Ablk = A + ((int64_t)kblk * NB) * ld + (int64_t)kblk * NB;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    STAMP(0);
    load_block_chol0(D, invd, &s_bad, Ablk, ld, tid, Xs);
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
        if (s_bad != 0 && *info == 0) *info = col_off + kblk * NB + s_bad;
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
    if (dinvB_all) {
        double* dB = dinvB_all + ((int64_t)by * nb + kblk) * (NB * NB);
        for (int e = tid; e < NB * NB / 2; e += NTH) {
            const int t = e >> 10, s2 = (e >> 6) & 15, l = e & 63;
            const int r = 16 * t + (l & 15), c = 8 * s2 + (l >> 4);
            d2 v;
            v[0] = (c <= r) ? D[r * LDD + c] : 0.0;
            v[1] = (c + 4 <= r) ? D[r * LDD + c + 4] : 0.0;
            *reinterpret_cast<d2*>(dB + 2 * e) = v;
        }
    }
    STAMP(5);
}

