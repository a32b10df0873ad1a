// chol16lp.hpp -- 16x16 Cholesky + inverse by ONE wave, lane-parallel (round 5).
//
// Replaces the MFMA / wave-uniform-scalar formulation (blocklds.hpp: chol16, four dependent 4x4 scalar blocks with two
// 8-pass MFMAs each: 350 cycles per pivot) on the critical path of the diagonal-block factorisation (the diagonal steps
// of torch.linalg.cholesky, reference call sites gpim/gpreg/gpr.py:192,248).
//
// Lane (q, i) = 16 q + i holds ROW i of the tile (all four 16-lane rows q hold the same data; a DPP row broadcast --
// row_newbcast:k, the one DPP control gfx950 allows on 64-bit operands -- reads lane k of the own 16-lane row).
// Square-root-free elimination A = Lt D Lt^T, one pivot per step j:
//     d_j = dg of lane j                            (one DPP broadcast; every lane keeps its own diagonal entry in dg)
//     inv_j = 1 / d_j                               (v_rcp_f64 + one third-order correction, uniform on all lanes)
//     c_i = a_ij inv_j  for i > j, 0 for i <= j     (lane i)
//     dg_i -= c_i a_ij                              (plain FMA)
//     a_ik -= c_i a_kj   (j < k < 15)               (one v_fmac_f64_dpp each: a_kj is register j of lane k)
//     m_ik -= c_i m_jk   (k <= j)                   (the same row operations on the identity: M = Lt^-1; 16-lane row q
//                                                    carries the columns k = q, q + 4, q + 8, q + 12 only)
// and afterwards, lane-parallel:  rs_i = dg_i^-1/2,  L_ij = a_ij^(j) rs_j,  X = L^-1: X_ik = rs_i m_ik.
// A lone wave issues one instruction per ~5 cycles and a dependent fp64 result is usable after 10 (v_rcp_f64: 20), so
// the ~280 instructions of the 16 steps are emitted in a fixed, list-scheduled order (gen_chol16lp.py ->
// chol16lp_steps.inc): the chain of pivot j+1 interleaved with the updates of pivot j.
#pragma once
#include "common.hpp"
#include <utility>

// d^-1/2 to within an ulp
__device__ __forceinline__ double lp_rsqrt(double d) {
    // seed good to 2^-24 (tools/r5_lat_probe.hip); e = 1 - d y^2;  y (1 + e/2 + 3 e^2 / 8) leaves 5 e^3 / 16 < 2^-70
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    const double p = fma(0.375 * e, e, 0.5 * e);
    return fma(y, p, y);
}

// Tile: element (rr, cc) at D[Lay::in(rr, cc)], lower part valid (the strictly upper part is read but never used).
// Writes to D the lower triangle of L -- SCALED: L itself; otherwise column j still carries the factor L_jj
// (D[i][j] = L_ij L_jj, the eliminated entries as they stand: whoever exports the tile multiplies column j by
// invd_out[j]; nothing else reads a diagonal tile of L: potf2_body.hpp) -- X = L^-1 (full tile, zeros above the diagonal)
// row-major with leading dimension ldx to Xout, 1/L_jj to invd_out[0..15]; returns 0 or 1 + first non-positive pivot
// (wave-uniform).  The upper triangle of the tile in D is left undefined.
#define LP_DECL16(P) double P##0, P##1, P##2, P##3, P##4, P##5, P##6, P##7, P##8, P##9, P##10, P##11, P##12, P##13, P##14, P##15
template <class Lay, bool SCALED = true>
__device__ __forceinline__ int chol16_lp(double* D, double* invd_out, int lane, double* Xout, int ldx) {
    int i = lane & 15;
    const int q = lane >> 4;
    // (the caller runs this in a loop over diagonal tiles: everything that depends on the lane only -- eight swizzled LDS
    // addresses, the start values of M -- is loop-invariant, and kept in registers across the loop it pushes the kernel
    // over its register budget: the compiler then spills it and reloads it here, on the critical path.  Recomputed instead.)
    asm volatile("" : "+v"(i));
    LP_DECL16(A);
#define LP_LOAD2(K0, K1)                                                                  \
    {                                                                                     \
        const double2 v = *reinterpret_cast<const double2*>(D + Lay::in(i, K0));          \
        A##K0 = v.x;                                                                      \
        A##K1 = v.y;                                                                      \
    }
    LP_LOAD2(0, 1) LP_LOAD2(2, 3) LP_LOAD2(4, 5) LP_LOAD2(6, 7) LP_LOAD2(8, 9) LP_LOAD2(10, 11) LP_LOAD2(12, 13) LP_LOAD2(14, 15)
#undef LP_LOAD2
    double DG = D[Lay::in(i, i)];
    int MSK = -1;
    double M0 = (q == i) ? 1.0 : 0.0, M1 = (4 + q == i) ? 1.0 : 0.0, M2 = (8 + q == i) ? 1.0 : 0.0, M3 = (12 + q == i) ? 1.0 : 0.0;
    // (these are loop-invariant for the caller's loop over diagonal tiles: without the next line the compiler keeps them
    // in registers of their own and copies one into the accumulator of its first v_fmac_f64_dpp right in front of it --
    // a VALU write of a DPP source two instructions too close, which the stream of chol16lp_steps.inc cannot see)
    asm volatile("" : "+v"(M0), "+v"(M1), "+v"(M2), "+v"(M3), "+v"(DG), "+v"(MSK));
#include "chol16lp_steps.inc"
    (void)A15;
    const unsigned long long badmask = __builtin_amdgcn_ballot_w64(!(DG > 0.0)) & 0xFFFFull;
    const int bad = badmask ? (int)__builtin_ctzll(badmask) + 1 : 0;
    const double rs = lp_rsqrt(DG);
#define LP_STORE2(K0, K1)                                                                          \
    {                                                                                              \
        double2 v;                                                                                 \
        v.x = A##K0;      /* (lane k's A_k went through the same operations as its DG: equal bits) */ \
        v.y = (K1 == 15 && i == 15) ? DG : A##K1;                                                  \
        if (SCALED) {                                                                              \
            v.x *= __builtin_amdgcn_update_dpp(0.0, rs, 0x150 + K0, 0xF, 0xF, true);               \
            v.y *= __builtin_amdgcn_update_dpp(0.0, rs, 0x150 + K1, 0xF, 0xF, true);               \
        }                                                                                          \
        if (q == 0) *reinterpret_cast<double2*>(D + Lay::in(i, K0)) = v;                           \
    }
    LP_STORE2(0, 1) LP_STORE2(2, 3) LP_STORE2(4, 5) LP_STORE2(6, 7) LP_STORE2(8, 9) LP_STORE2(10, 11) LP_STORE2(12, 13) LP_STORE2(14, 15)
#undef LP_STORE2
    if (q == 0) invd_out[i] = rs;
    Xout[i * ldx + q] = rs * M0;
    Xout[i * ldx + 4 + q] = rs * M1;
    Xout[i * ldx + 8 + q] = rs * M2;
    Xout[i * ldx + 12 + q] = rs * M3;
    return bad;
}
