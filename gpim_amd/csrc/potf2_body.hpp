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
// dinv_all[kblk] <- inverse of the factored block (ld 128, zeros above the diagonal); with inv_to_A the inverse
// replaces the block of L in A instead (the caller inverts the whole factor: cholstep.hip, plan_inverse).
// logdet_out[kblk] = sum_i log L_ii.  info: 1 + first failing global column (set once).
// dinvB_all (optional): the same inverse once more in the B-operand order of the f64 MFMA for the product
// S = P * Dinv^T of the panel solve (cholstep.hip): dinvB[t][s2][lane][e] = Dinv[16 t + (lane & 15)][8 s2 + 4 e + (lane >> 4)],
// so that a wave fetches the fragments of two k-steps of one 16-column tile with ONE coalesced 16-byte load per lane.
// smem: POTF2_SMEM_DOUBLES doubles of LDS -- the 36 lower 16x16 tiles of the block, packed (blocklds.hpp: LayTri), the
// reciprocal pivots, two scratch tiles: 79 392 bytes, so that TWO workgroups of a launch with this role fit one CU.
#ifndef FI_BURST
#define FI_BURST 8
#endif
// tools only (FI_ROLE_VARIANTS): the schedules of tools/r6_role_variants.hpp, measured in round 6 and not kept
#ifdef FI_ROLE_VARIANTS
#include "../../tools/r6_role_variants.hpp"
#endif
#ifndef FI_LOOKAHEAD
#define FI_LOOKAHEAD 0
#endif
#define POTF2_SMEM_DOUBLES (LayTri::DOUBLES + NB + 32 * XS_LD + 4)
typedef LayTri PL;
// LA (tools only): the look-ahead schedule of tools/r6_role_variants.hpp
template <typename R, bool LA = false>
__device__ __forceinline__ void potf2_body(double* __restrict__ smem, const int by, R* __restrict__ A, int64_t ld, int kblk,
                                           R* __restrict__ dinv_all, double* __restrict__ dinvB_all,
                                           double* __restrict__ logdet_out, int32_t* __restrict__ info, int nb PROF_ARG,
                                           int col_off = 0, bool inv_to_A = false) {
    double* D = smem;
    double* invd = D + PL::DOUBLES;
    double* Xs = invd + NB;
    double* red = Xs + 32 * XS_LD;
    int& s_bad = *reinterpret_cast<int*>(red + 2);
#ifdef FI_ROLE_VARIANTS
    int* fi_flags = reinterpret_cast<int*>(red + 2) + 1;        // three ints: the flags of lds_factor_inv_la
#endif
    A += (int64_t)by * nb * NB * ld;
    dinv_all += (int64_t)by * nb * NB * NB;
    logdet_out += (int64_t)by * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef R RV2 __attribute__((ext_vector_type(2)));
    R* Ablk = A + ((int64_t)kblk * NB) * ld + (int64_t)kblk * NB;
    if (tid == 0) s_bad = 0;
#ifdef FI_ROLE_VARIANTS
    if (tid < 3) fi_flags[tid] = 0;
#endif
    __syncthreads();
    STAMP(0);
    load_block_chol0(D, invd, &s_bad, Ablk, ld, tid, Xs);
    STAMP(1);
    // factor and invert in one sweep; block row i of L goes back to HBM (zeros above the diagonal)
    // during step i, just before the inverse overwrites it in LDS
    // 16-byte chunk e of block row i (16 rows x 128 columns) in tile-major order: tile e >> 7, row (e >> 3) & 15 of it,
    // column pair e & 7 -- one wave-wide access covers 8 rows x 128 bytes of ONE tile, which is conflict-free in the
    // packed layout (a matrix row across tiles would put all 64 lanes on the same 32 banks) and whole 128-byte lines
    // in HBM.  Elements above the diagonal are 0; tiles right of the diagonal tile do not exist in LDS.
    // factor_diag: the diagonal tile holds the factor as chol16_lp<.., false> leaves it, column j still times L_jj
    auto lower_chunk = [&](int i, int e, int& r, int& c, bool factor_diag) {
        const int tj = e >> 7, rr = (e >> 3) & 15, c2 = (e & 7) * 2;
        r = i * 16 + rr;
        c = tj * 16 + c2;
        d2 v = (d2){0.0, 0.0};
        if (tj <= i) {
            // (the tiles of the INVERSE are stored transposed: blocklds.hpp, Lay::XT)
            if (factor_diag || !PL::XT) v = *reinterpret_cast<const d2*>(D + PL::tile(i, tj) + PL::in(rr, c2));
            else v = (d2){D[PL::tile(i, tj) + PL::in(c2, rr)], D[PL::tile(i, tj) + PL::in(c2 + 1, rr)]};
        }
        if (factor_diag && tj == i) {
            v[0] *= invd[c];
            v[1] *= invd[c + 1];
        }
        if (c > r) v[0] = 0.0;
        if (c + 1 > r) v[1] = 0.0;
        return v;
    };
    // how the factor leaves (blocklds.hpp: lds_factor_inv)
    struct Sink {
        R* Ablk;
        int64_t ld;
        const double* D;
        const double* invd;
        // tile (t, p) from the solving wave's accumulator (blocklds.hpp: fi_frag order -- two column pairs of row lane & 15):
        // a store instruction covers 16 rows x 64 bytes
        __device__ __forceinline__ void tile(int t, int p, d4 acc, int lane) const {
            R* dst = Ablk + (int64_t)(16 * t + (lane & 15)) * ld + 16 * p + 2 * (lane >> 4);
            RV2 v;
            v[0] = (R)acc[0];
            v[1] = (R)acc[1];
            *reinterpret_cast<RV2*>(dst) = v;
            v[0] = (R)acc[2];
            v[1] = (R)acc[3];
            *reinterpret_cast<RV2*>(dst + 8) = v;
        }
        // tile (i, i-1), the diagonal tile (scaled, zeros above the diagonal) and the zero tiles right of it: one 16-byte
        // chunk per lane and tile (row lane >> 3 and row 8 + (lane >> 3), column pair lane & 7)
        __device__ __forceinline__ void row(int i, int lane) const {
            const int c2 = (lane & 7) * 2;
            d2 w[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = (lane >> 3) + 8 * h;
                w[0][h] = (i > 0) ? *reinterpret_cast<const d2*>(D + PL::tile(i, i - 1) + PL::in(rr, c2)) : (d2){0.0, 0.0};
                d2 v = *reinterpret_cast<const d2*>(D + PL::tile(i, i) + PL::in(rr, c2));
                v[0] = (c2 > rr) ? 0.0 : v[0] * invd[16 * i + c2];
                v[1] = (c2 + 1 > rr) ? 0.0 : v[1] * invd[16 * i + c2 + 1];
                w[1][h] = v;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                R* dst = Ablk + (int64_t)(16 * i + (lane >> 3) + 8 * h) * ld + c2;
                RV2 v;
                if (i > 0) {
                    v[0] = (R)w[0][h][0];
                    v[1] = (R)w[0][h][1];
                    *reinterpret_cast<RV2*>(dst + 16 * (i - 1)) = v;
                }
                v[0] = (R)w[1][h][0];
                v[1] = (R)w[1][h][1];
                *reinterpret_cast<RV2*>(dst + 16 * i) = v;
                v[0] = (R)0.0;
                v[1] = (R)0.0;
                for (int tj = i + 1; tj < 8; ++tj) *reinterpret_cast<RV2*>(dst + 16 * tj) = v;
            }
        }
    };
    const Sink sink{Ablk, ld, D, invd};
#ifdef FI_ROLE_VARIANTS
    if (LA) lds_factor_inv_la<Sink>(D, invd, Xs, &s_bad, fi_flags, tid, sink);
    else
#endif
    lds_factor_inv<Sink, true, PL>(D, invd, Xs, 8, &s_bad, tid, sink);
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
    // the inverse of the block: into dinv_all[kblk], or (fused inverse: a leaf of L^-1) over the block itself
    R* dinv = inv_to_A ? Ablk : dinv_all + (int64_t)kblk * NB * NB;
    const int64_t ldinv = inv_to_A ? ld : NB;
    for (int e = tid; e < NB * NB / 2; e += NTH) {
        int r, c;
        const d2 w = lower_chunk(e >> 10, e & 1023, r, c, false);
        RV2 v;
        v[0] = (R)w[0];
        v[1] = (R)w[1];
        *reinterpret_cast<RV2*>(dinv + r * ldinv + c) = v;
    }
    if (dinvB_all) {
        double* dB = dinvB_all + ((int64_t)by * nb + kblk) * (NB * NB);
        for (int e = tid; e < NB * NB / 2; e += NTH) {
            const int t = e >> 10, s2 = (e >> 6) & 15, l = e & 63;
            const int r = 16 * t + (l & 15), c = 8 * s2 + (l >> 4);     // c and c + 4 lie in the same 16-column tile
            const double* T = D + PL::tile(t, (s2 >> 1) <= t ? (s2 >> 1) : t);
            d2 v;
            v[0] = (c <= r) ? T[PL::XT ? PL::in(c & 15, r & 15) : PL::in(r & 15, c & 15)] : 0.0;
            v[1] = (c + 4 <= r) ? T[PL::XT ? PL::in((c + 4) & 15, r & 15) : PL::in(r & 15, (c + 4) & 15)] : 0.0;
            *reinterpret_cast<d2*>(dB + 2 * e) = v;
        }
    }
    STAMP(5);
}

