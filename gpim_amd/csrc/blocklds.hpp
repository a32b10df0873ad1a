// blocklds.hpp -- device helpers for a (<=128)x(<=128) lower-triangular block resident in LDS:
// 16x16 Cholesky / triangular inverse by one wave on MFMA, tile access, and the panel-wise
// factorisation with the row-wise inverse built from them.  Shared by potf2.hip (diagonal steps of
// the blocked Cholesky) and smalln.hip (fused trainer for N <= 128).  512-thread workgroups.
#pragma once
#include "common.hpp"
#include "chol16lp.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define LDD 130   // row stride of the LDS block: 130 % 32 == 2 keeps MK fragment reads conflict free
#define NTH 512
#define XS_LD 18   // row stride of the scratch tiles that hold the inverse of a diagonal 16x16 tile (lds_factor_inv)

// Where element (16 ti + rr, 16 tj + cc) of the block lives in LDS: D[Lay::tile(ti, tj) + Lay::in(rr, cc)].
//   LayPad  the full square, row-major with stride LDD (133 KB for 128 x 128): the fused small-N trainer, whose other
//           phases address D[i * LDD + j] directly.
//   LayTri  only the 36 lower 16x16 tiles of a 128 x 128 block, packed (tile (ti, tj) is number ti (ti + 1) / 2 + tj,
//           2 KB each = 72 KB): the diagonal-block role of the Cholesky step kernel -- under 80 KB with its scratch, so
//           two workgroups of that launch share a CU and the trailing-update tiles it hosts run two per CU
//           like in a launch of their own.  Rows of a tile are 16 doubles = 32 banks apart, so columns are XOR-swizzled
//           with 2 (rr >> 1): a ds_read_b64 is served in two groups of 32 lanes over 64 banks, and for all three
//           access patterns of this file -- A operand (rr = lane & 15, cc = 4 s + (lane >> 4)), B operand / accumulator
//           (rr = 4 s + (lane >> 4), cc = lane & 15) -- the 32 lanes of a group then hit 32 distinct bank pairs.  The
//           swizzle is even: the pair (cc, cc + 1), cc even, stays one aligned 16-byte chunk.
struct LayPad {
    static constexpr bool XT = false;      // lds_factor_inv leaves the tiles of the inverse as they are
    static constexpr int DOUBLES = NB * LDD;
    __device__ __forceinline__ static int tile(int ti, int tj) { return ti * 16 * LDD + tj * 16; }
    __device__ __forceinline__ static int in(int rr, int cc) { return rr * LDD + cc; }
};
struct LayTri {
    static constexpr bool XT = true;       // lds_factor_inv leaves every 16x16 tile of the inverse TRANSPOSED (see there)
    static constexpr int DOUBLES = 36 * 256;
    __device__ __forceinline__ static int tile(int ti, int tj) { return ((ti * (ti + 1)) / 2 + tj) * 256; }
    __device__ __forceinline__ static int in(int rr, int cc) { return rr * 16 + (cc ^ ((rr >> 1) << 1)); }
};

__device__ __forceinline__ double bcast_lane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// DPP lane permutation inside 16-lane rows (0xB1 / 0x4E: quad_perm xor 1 / xor 2, 0x141: row_half_mirror,
// 0x140: row_mirror) -- VALU moves, no LDS crossbar round trip as with ds_bpermute (__shfl_xor)
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v;
}
// sum over the 64 lanes, wave-uniform result, fixed order
__device__ __forceinline__ double wave_sum(double v) {
    v = quad_sum(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return (bcast_lane(v, 0) + bcast_lane(v, 16)) + (bcast_lane(v, 32) + bcast_lane(v, 48));
}

// sqrt(d) and 1/sqrt(d): v_rsq_f64 seed + two Newton steps on both (a third of the dependent
// instruction chain of sqrt() followed by a divide)
__device__ __forceinline__ void sqrt_rsqrt(double d, double& l, double& inv) {
    inv = __builtin_amdgcn_rsq(d);
    l = d * inv;
    l = fma(0.5 * inv, fma(-l, l, d), l);
    inv = fma(inv, fma(-l, inv, 1.0), inv);
    l = fma(0.5 * inv, fma(-l, l, d), l);
    inv = fma(inv, fma(-l, inv, 1.0), inv);
}

__device__ __forceinline__ double pick4(double v0, double v1, double v2, double v3, int i) {
    double r = v0;
    r = (i == 1) ? v1 : r;
    r = (i == 2) ? v2 : r;
    r = (i == 3) ? v3 : r;
    return r;
}

// 16x16 lower Cholesky by one full wave; D points at the tile's origin (element (rr, cc) at D[Lay::in(rr, cc)], lower
// part valid).  Writes L16 (lower) back and 1/L_jj to invd_out[0..15]; returns 0 or 1 + first bad
// local column (wave-uniform).
//
// The tile lives in the MFMA accumulator layout of the symmetric matrix: lane l, register g holds
// A[l&15][4g + (l>>4)], so register g *is* the 16x4 column panel g in f64 16x16x4 operand layout.
// Per panel: the 4x4 diagonal block is factored and inverted in wave-uniform scalars (10 readlanes),
// one MFMA forms the panel  L_p^T = W A_p^T  (W = L4^-1) and one MFMA applies  A -= L_p L_p^T.
//
// xinv (optional): the inverse X = L16^-1 as a by-product, in accumulator layout X[(lane>>4) + 4g][lane & 15] (what
// trinv16_regs returns).  The same block elimination applied to the identity: with Y = I, per 4-column panel
// X[rows of the panel] = W Y[rows of the panel] and Y -= L_p X[rows of the panel] -- two more MFMAs per panel that are
// independent of the factorisation's own dependency chain.  Cost, one wave alone (tools/chol16_var.hip): 4.43K cycles
// without, 5.23K with the inverse (fp64 MFMAs and the wave's fp64 scalar chains share the SIMD's fp64 unit); handing
// the two operands per panel to a helper wave through LDS behind a flag instead measured 5.60K -- not kept.
template <class Lay = LayPad>
__device__ __forceinline__ int chol16(double* D, double* invd_out, int lane, d4* xinv = nullptr) {
    const int r = lane & 15, kq = lane >> 4, p = r & 3;
    d4 acc, Lf;
    d4 yacc, Xf;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        yacc[g] = ((kq + 4 * g) == r) ? 1.0 : 0.0;
        Xf[g] = 0.0;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = kq + 4 * g;
        acc[g] = D[Lay::in(r > c ? r : c, r > c ? c : r)];
    }
    int bad = 0;
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        const double P = acc[jb];
        const int b0 = 4 * jb;
        // A'[b0+i][b0+q] sits in lane (b0+i) + 16 q
        const double a00 = bcast_lane(P, b0), a10 = bcast_lane(P, b0 + 1), a20 = bcast_lane(P, b0 + 2),
                     a30 = bcast_lane(P, b0 + 3), a11 = bcast_lane(P, b0 + 17), a21 = bcast_lane(P, b0 + 18),
                     a31 = bcast_lane(P, b0 + 19), a22 = bcast_lane(P, b0 + 34), a32 = bcast_lane(P, b0 + 35),
                     a33 = bcast_lane(P, b0 + 51);
        double l00, l11, l22, l33, i0, i1, i2, i3;
        sqrt_rsqrt(a00, l00, i0);
        const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
        const double d1 = fma(-l10, l10, a11);
        sqrt_rsqrt(d1, l11, i1);
        const double l21 = fma(-l20, l10, a21) * i1, l31 = fma(-l30, l10, a31) * i1;
        const double d2 = fma(-l21, l21, fma(-l20, l20, a22));
        sqrt_rsqrt(d2, l22, i2);
        const double l32 = fma(-l31, l21, fma(-l30, l20, a32)) * i2;
        const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a33)));
        sqrt_rsqrt(d3, l33, i3);
        if (bad == 0) {
            if (!(a00 > 0.0)) bad = b0 + 1;
            else if (!(d1 > 0.0)) bad = b0 + 2;
            else if (!(d2 > 0.0)) bad = b0 + 3;
            else if (!(d3 > 0.0)) bad = b0 + 4;
        }
        // W = L4^-1
        const double w10 = -i1 * (l10 * i0), w21 = -i2 * (l21 * i1), w32 = -i3 * (l32 * i2);
        const double w20 = -i2 * fma(l21, w10, l20 * i0), w31 = -i3 * fma(l32, w21, l31 * i1);
        const double w30 = -i3 * fma(l32, w20, fma(l31, w10, l30 * i0));
        // operand A of the panel MFMA: rows m = r < 4 of W, column k = kq
        const double wsel = pick4(pick4(i0, w10, w20, w30, p), pick4(0.0, i1, w21, w31, p),
                                  pick4(0.0, 0.0, i2, w32, p), (p == 3) ? i3 : 0.0, kq);
        const double wop = (r < 4) ? wsel : 0.0;
        const d4 res = __builtin_amdgcn_mfma_f64_16x16x4f64(wop, P, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        // (issued next to `res`, which it does not depend on: both are in flight while the panel is assembled)
        d4 xr = (d4){0.0, 0.0, 0.0, 0.0};
        if (xinv) xr = __builtin_amdgcn_mfma_f64_16x16x4f64(wop, yacc[jb], (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        // rows of the diagonal block take the scalar factor itself; rows above it are zero
        const double lsel = pick4(pick4(l00, l10, l20, l30, p), pick4(0.0, l11, l21, l31, p),
                                  pick4(0.0, 0.0, l22, l32, p), (p == 3) ? l33 : 0.0, kq);
        double Lp = res[0];
        Lp = ((r >> 2) == jb) ? lsel : Lp;
        Lp = ((r >> 2) < jb) ? 0.0 : Lp;
        Lf[jb] = Lp;
        if (jb < 3) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lp, Lp, acc, 0, 0, 0);
        if (xinv) {
            Xf[jb] = xr[0];
            if (jb < 3) yacc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lp, xr[0], yacc, 0, 0, 0);
        }
        if (lane == 0) {
            invd_out[b0] = i0;
            invd_out[b0 + 1] = i1;
            invd_out[b0 + 2] = i2;
            invd_out[b0 + 3] = i3;
        }
    }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb) {
        const int c = 4 * jb + kq;
        if (c <= r) D[Lay::in(r, c)] = Lf[jb];
    }
    if (xinv) *xinv = Xf;
    return bad;
}

// row-major copy of a 16x16 tile held in accumulator layout (scratch tile of lds_factor_inv: the layout converter
// between "accumulator" and "operand" order)
__device__ __forceinline__ void xs_write(double* Xs, d4 x, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) Xs[((lane >> 4) + 4 * g) * 18 + (lane & 15)] = x[g];
}

// Inverse of a 16x16 lower-triangular block by one full wave: X = L^-1 written as a full tile
// (zeros above the diagonal).  `out` may alias `L`.
//
// Recursive doubling on MFMA.  R(M)[s] = M[l&15][4s+(l>>4)] is the A-operand layout of k-step s,
// C(M)[g] = M[(l>>4)+4g][l&15] the B-operand / accumulator layout; R(M) == C(M^T).  Level 0: every
// lane back-substitutes the two columns of its own 4x4 diagonal block it needs.  Each level then is
// X <- X - X (M X) with M the sub-diagonal blocks absorbed at that level; level 1 is also run
// transposed so that level 2 has X in both layouts.
__device__ __forceinline__ d4 trinv16_regs(const double* L, int ldl, const double* invd, int lane) {
    const int r = lane & 15, kq = lane >> 4, cb = r >> 2, p = r & 3;
    const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
    d4 RL, CL;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 4 * s + kq;
        RL[s] = (c <= r) ? L[r * ldl + c] : 0.0;       // L[r][4s+kq]
        CL[s] = (r <= c) ? L[c * ldl + r] : 0.0;       // L[4s+kq][r]
    }
    const double* Lb = L + (4 * cb) * ldl + 4 * cb;
    const double l10 = Lb[ldl], l20 = Lb[2 * ldl], l21 = Lb[2 * ldl + 1];
    const double l30 = Lb[3 * ldl], l31 = Lb[3 * ldl + 1], l32 = Lb[3 * ldl + 2];
    const double i0 = invd[4 * cb], i1 = invd[4 * cb + 1], i2 = invd[4 * cb + 2], i3 = invd[4 * cb + 3];
    double valC, valR;
    {   // column p of W_cb, row kq  -> C layout;  column kq, row p -> R layout
        const double x0 = (p == 0) ? i0 : 0.0;
        const double x1 = fma(-l10, x0, (p == 1) ? 1.0 : 0.0) * i1;
        const double x2 = fma(-l21, x1, fma(-l20, x0, (p == 2) ? 1.0 : 0.0)) * i2;
        const double x3 = fma(-l32, x2, fma(-l31, x1, fma(-l30, x0, (p == 3) ? 1.0 : 0.0))) * i3;
        valC = pick4(x0, x1, x2, x3, kq);
        const double y0 = (kq == 0) ? i0 : 0.0;
        const double y1 = fma(-l10, y0, (kq == 1) ? 1.0 : 0.0) * i1;
        const double y2 = fma(-l21, y1, fma(-l20, y0, (kq == 2) ? 1.0 : 0.0)) * i2;
        const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, (kq == 3) ? 1.0 : 0.0))) * i3;
        valR = pick4(y0, y1, y2, y3, p);
    }
    d4 XC, XR;      // C(X0), R(X0)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        XC[g] = (g == cb) ? valC : 0.0;
        XR[g] = (g == cb) ? valR : 0.0;
    }
    // level 1: M1 = blocks (1,0), (3,2)
    d4 X1, X1T;
    {
        d4 T = zero;
#pragma unroll
        for (int s = 0; s < 4; s += 2)      // M1 has column blocks 0, 2
            T = __builtin_amdgcn_mfma_f64_16x16x4f64(((cb & 1) && s == cb - 1) ? RL[s] : 0.0, XC[s], T, 0, 0, 0);
        X1 = XC;
#pragma unroll
        for (int s = 1; s < 4; s += 2)      // T has row blocks 1, 3
            X1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-XR[s], T[s], X1, 0, 0, 0);
        d4 Tt = zero;                       // M1^T X0^T: M1^T has column blocks 1, 3
#pragma unroll
        for (int s = 1; s < 4; s += 2)
            Tt = __builtin_amdgcn_mfma_f64_16x16x4f64((cb == s - 1) ? CL[s] : 0.0, XR[s], Tt, 0, 0, 0);
        X1T = XR;
#pragma unroll
        for (int s = 0; s < 4; s += 2)      // Tt has row blocks 0, 2
            X1T = __builtin_amdgcn_mfma_f64_16x16x4f64(-XC[s], Tt[s], X1T, 0, 0, 0);
    }
    // level 2: M2 = rows 8..15 x columns 0..7
    d4 X2 = X1;
    {
        d4 T = zero;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            T = __builtin_amdgcn_mfma_f64_16x16x4f64((cb >= 2) ? RL[s] : 0.0, X1[s], T, 0, 0, 0);
#pragma unroll
        for (int s = 2; s < 4; ++s)
            X2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-X1T[s], T[s], X2, 0, 0, 0);
    }
    return X2;      // accumulator layout: X[(lane>>4) + 4g][lane & 15]
}
__device__ __forceinline__ void trinv16(const double* L, int ldl, const double* invd, double* out, int ldo,
                                        int lane) {
    const d4 X = trinv16_regs(L, ldl, invd, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) out[((lane >> 4) + 4 * g) * ldo + (lane & 15)] = X[g];
}

template <class Lay = LayPad>
__device__ __forceinline__ d4 tile_read(const double* C, int lane) {
    d4 v;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) v[rg] = C[Lay::in((lane >> 4) + 4 * rg, lane & 15)];
    return v;
}
template <class Lay = LayPad>
__device__ __forceinline__ void tile_write(double* C, d4 v, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) C[Lay::in((lane >> 4) + 4 * rg, lane & 15)] = v[rg];
}

// Loads the lower 16x16 tiles of the 128x128 block into D (LayTri) AND factors its first tile: all 9 loads of a
// thread are put in flight (16-byte chunk q = tid + 512 i of the packed lower triangle: tile q >> 7 in the order
// (0,0) (1,0) (1,1) (2,0) ..., row (q >> 3) & 15 of it, column pair q & 7 -- eight lanes fetch one 128-byte row segment),
// the first chunk of every thread (tiles 0 - 3) is stored as soon as it arrives, and wave 0 runs chol16 on tile (0,0)
// while the rest of the block is still on its way.  Pairs with lds_factor_inv<.., true, LayTri>.  The strictly upper
// tiles of the block are never read.
// R: element type of the matrix in HBM (double, or float for single-precision handles; the block is factored in
// double either way).
__device__ __forceinline__ void tri_tile_of(int t, int& ti, int& tj) {
    ti = (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15) + (t >= 21) + (t >= 28);
    tj = t - (ti * (ti + 1)) / 2;
}
template <typename R>
__device__ __forceinline__ void load_block_chol0(double* D, double* invd, int* s_bad,
                                                 const R* __restrict__ Ablk, int64_t ld, int tid, double* Xs) {
    typedef R RV2 __attribute__((ext_vector_type(2)));
    constexpr int NCH = 36 * 128 / NTH;      // 9
    d2 r[NCH];
    int off[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int q = tid + NTH * i, rr = (q >> 3) & 15, c2 = (q & 7) * 2;
        int ti, tj;
        tri_tile_of(q >> 7, ti, tj);
        const RV2 v = *reinterpret_cast<const RV2*>(Ablk + (int64_t)(ti * 16 + rr) * ld + tj * 16 + c2);
        r[i] = (d2){(double)v[0], (double)v[1]};
        off[i] = (q >> 7) * 256 + LayTri::in(rr, c2);
    }
    *reinterpret_cast<d2*>(D + off[0]) = r[0];
    __syncthreads();
    if (tid < 64) {
        const int bad = chol16_lp<LayTri, false>(D, invd, tid, Xs, XS_LD);
        if (tid == 0 && bad && *s_bad == 0) *s_bad = bad;
    }
#pragma unroll
    for (int i = 1; i < NCH; ++i) *reinterpret_cast<d2*>(D + off[i]) = r[i];
    __syncthreads();
}


// Cholesky AND inverse of the leading npan*16 rows/cols of the block in D, in place: on return D holds
// X = L^-1 (lower; the diagonal 16x16 tiles with zeros above the diagonal), invd[j] = 1/L_jj.
// Right-looking factorisation in 16-column panels.  The serial part is wave 0: own tile update -> 16x16 factorisation
// of the next diagonal tile by ONE wave (chol16_lp, which also yields that tile's INVERSE; 2.3K cycles) -- overlapped
// with the trailing update by the other waves.  The panel solve of the rows below a diagonal tile is a product with that
// inverse, one 16x16 tile per wave on MFMA.  The inverse of the whole block is built block row by block row in the
// shadow of the factorisation instead of afterwards:
//   X(i,i) = L(i,i)^-1,   X(i,j) = -X(i,i) * sum_{k=j}^{i-1} L(i,k) X(k,j)      (j < i)
// Block row i of L is final once step i-1 is over.  During step i
//   panel-solve phase : every wave but wave 4 solves one tile below the diagonal tile (and hands it to sink.tile() from its
//                       registers); wave 0 solves its tile TRANSPOSED -- X(i,i) A^T lands in the accumulator as the
//                       A-operand fragments of the solved tile -- and updates the next diagonal tile with it at once,
//                       without a round trip through LDS; wave 4 stores X(i-1,i-1) (from the scratch tile) over
//                       L(i-1,i-1); the workers store the tiles of block row i-1 of X they hold in registers since the
//                       previous step over L(i-1,.);
//   update phase      : wave 4 (the SIMD partner of wave 0: no matrix work beside the serial chain) exports what is left
//                       of block row i of L (sink.row(i, lane)); the six workers (waves 1-3, 5-7) share the trailing tiles and the tiles
//                       of row i of the inverse -- T(i,j) = sum_k L(i,k) X(k,j), times -X(i,i) -- by a static
//                       longest-first assignment (FiPlan below); the inverse's results stay in registers until the next
//                       step so that no wave overwrites an L(i,k) another one still reads.  The inverse's work grows as
//                       the trailing update shrinks (88 ... 140 MFMAs per step for eight panels), and both hide behind
//                       wave 0's 16x16 factorisation.  In the LAST step wave 0 has no factorisation left and works too.
// An fp64 MFMA occupies its SIMD's matrix pipe for 64 cycles -- which is also when a dependent one can issue
// (tools/r5_lat_probe.hip, tools/mfma_peak.hip) -- so one accumulation chain per tile product is as fast as two, and what
// the update phase costs is its MFMA count per SIMD: 896 MFMAs in all for eight panels.
// Xs: TWO scratch tiles (2 x 16 x XS_LD doubles): the inverse of diagonal tile i lives in Xs + (i & 1) * 16 * XS_LD,
// row-major, from the end of step i-1 to the panel-solve phase of step i+1.
#ifndef FI_EXP
#define FI_EXP 0
#endif
#ifdef POTF2_PROFILE
__device__ long long g_fprof[128];       // tools/potf2_prof.hip: wave 0's clock at the marks below, 8 per 16-column step
#define FSTAMP(i) do { if (threadIdx.x == 0) g_fprof[i] = clock64(); } while (0)
__device__ long long g_wprof[8 * 8 * 4];  // every wave's clock: [wave][step][after export | after row of the inverse | after trailing tiles | after the step's barrier]
#define WSTAMP(p, k) do { if ((threadIdx.x & 63) == 0) g_wprof[((threadIdx.x >> 6) * 8 + (p)) * 4 + (k)] = clock64(); } while (0)
#else
#define FSTAMP(i) do { } while (0)
#define WSTAMP(p, k) do { } while (0)
#endif
// The factor leaves through `sink` (the inverse overwrites it in LDS):
//   sink.tile(t, p, acc, lane)  tile (t, p), t > p + 1, from the registers of the wave that solved it, in the panel-solve
//                               phase of step p: acc = fi_frag of the tile, i.e. acc[2 h + e] = L[16 t + (lane & 15)][16 p + 8 h + 2 (lane >> 4) + e];
//   sink.row(i, lane)           by wave 4 in the update phase of step i: the rest of block row i -- tile (i, i-1) (solved
//                               by wave 0, which has no time for stores), the diagonal tile (column j still times L_jj:
//                               chol16lp.hpp) and the zeros right of it.
// (Round 3: all 512 threads exported whole block rows in the panel-solve phase, on wave 0's critical path; round 4: the six
// workers, 1.0-1.2K cycles each in front of their tiles; one wave alone needs 3.6K per row.)
struct FiNoSink {
    __device__ __forceinline__ void tile(int, int, d4, int) const {}
    __device__ __forceinline__ void row(int, int) const {}
};

// Who does what in the update phase of step p of an npan-panel block: w[npan][p][worker] packs
//   bits 0-3 / 4-7   the tiles j of row p of the inverse this worker forms (15 = none; the first is the smaller j),
//   bits 8-11        its number of trailing ITEMS, then 12 bits each: (rtA << 9) | (ctA << 6) | (rtB << 3) | ctB --
//                    two tiles updated together (rtB = 0: one tile alone).
// Workers 0-5 are waves 1, 2, 3, 5, 6, 7; worker 6 is wave 0, which takes part in the last step only.  An fp64 MFMA
// occupies its SIMD for 64 cycles whichever wave issued it, so the load is balanced per SIMD first (workers w and w + 3
// share SIMD w + 1; worker 6 has SIMD 0 to itself) and between the two waves of a SIMD second: longest item first; an
// inverse tile costs its k-range + 1 (the product with -X(p,p)), a trailing tile 1.
// Why pairs (tools/r5_trail_probe.hip, cycles per tile and SIMD with two waves per SIMD; the MFMAs alone: 261): every
// LDS instruction costs the SIMD ~20 cycles of MFMA issue whichever wave it comes from, and a tile alone is a chain of
// load latency -> four dependent MFMAs -> store.  One tile per loop trip with the next one's operands prefetched
// (round 5's first version: copies between register sets, each stalling on the MFMA that still reads its target) 669,
// one tile per trip without prefetch 508, two tiles of one block row per trip (shared A fragments, two accumulator
// chains) 348, two unrelated tiles 379.
struct FiPlan {
    unsigned long long w[9][8][7];
};
constexpr FiPlan make_fi_plan() {
    FiPlan P{};
    for (int npan = 1; npan <= 8; ++npan)
        for (int p = 0; p < npan; ++p) {
            const bool lastp = (p == npan - 1);
            int load[7] = {0, 0, 0, 0, 0, 0, 0}, ninv[7] = {0, 0, 0, 0, 0, 0, 0}, ntr[7] = {0, 0, 0, 0, 0, 0, 0};
            unsigned long long word[7] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
            auto pick = [&](bool inverse) {
                // least-loaded SIMD, then its less loaded wave (an inverse tile: a wave that holds fewer than two)
                int best = -1, best_simd = 0, best_wave = 0;
                for (int w = 0; w < (lastp ? 7 : 6); ++w) {
                    if (inverse && ninv[w] >= 2) continue;
                    const int simd = (w == 6) ? load[6] : load[w % 3] + load[w % 3 + 3];
                    if (best < 0 || simd < best_simd || (simd == best_simd && load[w] < best_wave)) {
                        best = w;
                        best_simd = simd;
                        best_wave = load[w];
                    }
                }
                return best;
            };
            for (int j = 0; j < p; ++j) {
                const int w = pick(true);
                word[w] = (word[w] & ~(0xFull << (4 * ninv[w]))) | ((unsigned long long)j << (4 * ninv[w]));
                ++ninv[w];
                load[w] += (p - j) + 1;
            }
            if (!lastp) {
                // the trailing tiles in row-major order, dealt in contiguous runs (neighbours in a run mostly share their block
                // row, hence their A fragments) whose lengths even out the load: a trailing tile weighs 4, one 16-deep
                // product of an inverse tile 3 (half the LDS instructions)
                int tiles[32] = {}, nt = 0, cnt[7] = {0, 0, 0, 0, 0, 0, 0}, wl[7] = {0, 0, 0, 0, 0, 0, 0};
                for (int rt = npan - 1; rt > p; --rt)
                    for (int ct = p + 1; ct <= rt; ++ct)
                        if (!(rt == p + 1 && ct == p + 1)) tiles[nt++] = (rt << 3) | ct;      // (not wave 0's own tile)
                for (int w = 0; w < 6; ++w) wl[w] = 3 * load[w];
                for (int i = 0; i < nt; ++i) {
                    int best = 0;
                    for (int w = 1; w < 6; ++w) {
                        const int sb = wl[best % 3] + wl[best % 3 + 3], sw = wl[w % 3] + wl[w % 3 + 3];
                        if (sw < sb || (sw == sb && wl[w] < wl[best])) best = w;
                    }
                    ++cnt[best];
                    wl[best] += 4;
                }
                int at = 0;
                for (int w = 0; w < 6; ++w) {
                    int i = 0;
                    for (; i + 1 < cnt[w]; i += 2) {
                        word[w] |= (unsigned long long)((tiles[at + i] << 6) | tiles[at + i + 1]) << (12 + 12 * ntr[w]);
                        ++ntr[w];
                    }
                    if (i < cnt[w]) {
                        word[w] |= (unsigned long long)(tiles[at + i] << 6) << (12 + 12 * ntr[w]);
                        ++ntr[w];
                    }
                    at += cnt[w];
                }
            }
            for (int w = 0; w < 7; ++w) P.w[npan][p][w] = word[w] | ((unsigned long long)ntr[w] << 8);
        }
    return P;
}
static __constant__ FiPlan c_fi_plan = make_fi_plan();

// Operand fragments with 16-byte LDS accesses.  The four k-steps of a 16-deep product may visit the 16 columns of the
// operand tiles in any order as long as both operands use the same one; with
//     kappa(s, kq) = 8 (s >> 1) + 2 kq + (s & 1)        (k-step s, lane group kq = lane >> 4)
// instead of 4 s + kq a lane's four values are two aligned pairs of ONE row -- columns 2 kq, 2 kq + 1 and 8 + 2 kq,
// 9 + 2 kq -- i.e. two ds_read_b128 instead of four ds_read_b64 (both layouts keep an even-aligned column pair in one
// 16-byte chunk; LayTri's swizzle stays conflict-free for the four 16-lane groups of a b128 access, also with the row
// permutation rho below).  An LDS instruction costs its SIMD ~10-25 cycles of MFMA issue whichever wave it comes from
// (tools/r5_trail_probe.hip: a 16-deep product with 8 ds_read_b64 423 cycles per SIMD, with 4 ds_read_b128 312, the four
// MFMAs alone 262), so the instruction count is what the update phase pays for.
//     fi_frag(T, row, kq)[s] = T[row][kappa(s, kq)]
template <class Lay>
__device__ __forceinline__ d4 fi_frag(const double* T, int row, int kq) {
    const d2 lo = *reinterpret_cast<const d2*>(T + Lay::in(row, 2 * kq)), hi = *reinterpret_cast<const d2*>(T + Lay::in(row, 8 + 2 * kq));
    return (d4){lo[0], lo[1], hi[0], hi[1]};
}
template <class Lay>
__device__ __forceinline__ void fi_frag_store(double* T, int row, int kq, d4 v) {
    *reinterpret_cast<d2*>(T + Lay::in(row, 2 * kq)) = (d2){v[0], v[1]};
    *reinterpret_cast<d2*>(T + Lay::in(row, 8 + 2 * kq)) = (d2){v[2], v[3]};
}
// The accumulator of the MFMA is fixed: lane (r, kq), register g holds Out[4 g + kq][r].  With the A operand's rows
// permuted by rho(i) = kappa(i >> 2, i & 3) the products below come out as Out[i][j] = C[j][rho(i)], so register g of
// lane (r, kq) is C[r][kappa(g, kq)]: the accumulator IS fi_frag of the result tile and moves with two 16-byte accesses.
__device__ __forceinline__ int fi_rho(int i) { return 8 * (i >> 3) + 2 * (i & 3) + ((i >> 2) & 1); }
// B-operand fragments of a tile used untransposed: f[s] = T[kappa(s, kq)][r]
template <class Lay>
__device__ __forceinline__ d4 fi_fragT(const double* T, int r, int kq) {
    return (d4){T[Lay::in(2 * kq, r)], T[Lay::in(2 * kq + 1, r)], T[Lay::in(8 + 2 * kq, r)], T[Lay::in(9 + 2 * kq, r)]};
}
// fragments of a 16x16 scratch tile (row-major, leading dimension XS_LD) in the same column order
__device__ __forceinline__ d4 fi_frag_xs(const double* Xp, int row, int kq) {
    const d2 lo = *reinterpret_cast<const d2*>(Xp + row * XS_LD + 2 * kq), hi = *reinterpret_cast<const d2*>(Xp + row * XS_LD + 8 + 2 * kq);
    return (d4){lo[0], lo[1], hi[0], hi[1]};
}
// C(rt, ct) -= L(rt, p) L(ct, p)^T for one tile: Out[i][j] = sum_c L(ct, p)[rho(i)][c] L(rt, p)[j][c]
template <class Lay>
__device__ __forceinline__ void fi_trail1(double* D, int p, int rt, int ct, int lane) {
    const int r = lane & 15, kq = lane >> 4;
    const d4 a = fi_frag<Lay>(D + Lay::tile(ct, p), fi_rho(r), kq), b = fi_frag<Lay>(D + Lay::tile(rt, p), r, kq);
    double* C = D + Lay::tile(rt, ct);
    d4 c = fi_frag<Lay>(C, r, kq);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[s4], b[s4], c, 0, 0, 0);
    fi_frag_store<Lay>(C, r, kq, c);
}
// ... for two tiles at once: two accumulator chains; SHARED: both in block row rtA (one set of fragments of L(rtA, p))
template <class Lay, bool SHARED>
__device__ __forceinline__ void fi_trail2(double* D, int p, int rtA, int ctA, int rtB, int ctB, int lane) {
    const int r = lane & 15, kq = lane >> 4, rr = fi_rho(r);
    const d4 b0 = fi_frag<Lay>(D + Lay::tile(rtA, p), r, kq);
    const d4 b1 = SHARED ? b0 : fi_frag<Lay>(D + Lay::tile(rtB, p), r, kq);
    const d4 a0 = fi_frag<Lay>(D + Lay::tile(ctA, p), rr, kq), a1 = fi_frag<Lay>(D + Lay::tile(ctB, p), rr, kq);
    double* C0 = D + Lay::tile(rtA, ctA);
    double* C1 = D + Lay::tile(rtB, ctB);
    d4 c0 = fi_frag<Lay>(C0, r, kq), c1 = fi_frag<Lay>(C1, r, kq);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[s4], b0[s4], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[s4], b1[s4], c1, 0, 0, 0);
    }
    fi_frag_store<Lay>(C0, r, kq, c0);
    fi_frag_store<Lay>(C1, r, kq, c1);
}
// X(p, j) = -X(p, p) * sum_{k = j}^{p-1} L(p, k) X(k, j) for the tiles j1 < j2 of one worker (j2 = 15: j1 alone): where both
// sums run they share the fragments of L(p, k) and their two accumulator chains alternate.
// Lay::XT: the tiles of the inverse live in D TRANSPOSED -- the contraction runs over the ROWS of X(k, j), and only a
// tile's columns come in 16-byte pairs: X(k, j)[kappa(s, kq)][r] = X(k, j)^T[r][kappa(s, kq)] is fi_frag of the stored
// tile, and with the rows of -X(p, p) permuted by rho the result leaves the accumulator as fi_frag of X(p, j)^T.
// Otherwise (the fused small-N trainer reads the inverse in place) the B operand is gathered with four 8-byte reads and the
// result x[g] = X(p, j)[4 g + kq][r] is an accumulator tile (tile_write).
template <class Lay>
__device__ __forceinline__ d4 fi_xfrag(const double* T, int r, int kq) {
    return Lay::XT ? fi_frag<Lay>(T, r, kq) : fi_fragT<Lay>(T, r, kq);
}
// the lane index, opaque to the optimiser: address arithmetic derived from it is redone where it is used instead of being
// hoisted out of the step loop and kept live across it (chol_step_kernel at 128 registers spilled two such offsets)
__device__ __forceinline__ int fi_lane_opaque() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
template <class Lay>
__device__ __forceinline__ void fi_inv_pair(const double* D, const double* Xp, int p, int j1, int j2, int lane_, d4& x1, d4& x2) {
    const int lane = fi_lane_opaque();
    (void)lane_;
    const int r = lane & 15, kq = lane >> 4;
    d4 xa;
    {
        const int row = Lay::XT ? fi_rho(r) : r;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) xa[s4] = -Xp[row * XS_LD + 4 * s4 + kq];
    }
    d4 t1 = (d4){0.0, 0.0, 0.0, 0.0}, t2 = t1;
    const int ke = (j2 != 15) ? j2 : p;
    int k = j1;
    // (two register sets taking turns: the operands of step k + 1 are in flight during the MFMAs of step k, and no set is
    // copied into another -- a copy waits for the MFMA that still reads its target)
    if (k < ke) {
        d4 a0 = fi_frag<Lay>(D + Lay::tile(p, k), r, kq), b0 = fi_xfrag<Lay>(D + Lay::tile(k, j1), r, kq), a1 = a0, b1 = b0;
        for (; k < ke; k += 2) {
            if (k + 1 < ke) {
                a1 = fi_frag<Lay>(D + Lay::tile(p, k + 1), r, kq);
                b1 = fi_xfrag<Lay>(D + Lay::tile(k + 1, j1), r, kq);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s4], b0[s4], t1, 0, 0, 0);
            if (k + 1 >= ke) break;
            if (k + 2 < ke) {
                a0 = fi_frag<Lay>(D + Lay::tile(p, k + 2), r, kq);
                b0 = fi_xfrag<Lay>(D + Lay::tile(k + 2, j1), r, kq);
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s4], b1[s4], t1, 0, 0, 0);
        }
        k = ke;
    }
    for (; k < p; ++k) {
        const d4 a = fi_frag<Lay>(D + Lay::tile(p, k), r, kq);
        const d4 b1 = fi_xfrag<Lay>(D + Lay::tile(k, j1), r, kq), b2 = fi_xfrag<Lay>(D + Lay::tile(k, j2), r, kq);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b1[s4], t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b2[s4], t2, 0, 0, 0);
        }
    }
    // (the B operand comes straight from the accumulators: t[g] = T[4 g + kq][r], so k-step g contracts rows 4 g + kq)
    d4 y1 = (d4){0.0, 0.0, 0.0, 0.0}, y2 = y1;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s4], t1[s4], y1, 0, 0, 0);
        y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s4], t2[s4], y2, 0, 0, 0);
    }
    x1 = y1;
    x2 = y2;
}
// a tile of the inverse from the accumulator of fi_inv_pair into D
template <class Lay>
__device__ __forceinline__ void fi_xstore(double* T, d4 x, int lane) {
    if (Lay::XT) fi_frag_store<Lay>(T, lane & 15, lane >> 4, x);
    else tile_write<Lay>(T, x, lane);
}

// FIRST_DONE: the caller has already factored the first diagonal tile (load_block_chol0).
template <typename Sink, bool FIRST_DONE = false, class Lay = LayPad>
__device__ __forceinline__ void lds_factor_inv(double* D, double* invd, double* Xs, int npan, int* s_bad, int tid,
                                               Sink sink) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: tile-base arithmetic on the SALU)
    const int r = lane & 15, kq = lane >> 4;
    const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
    const int wid = (wave == 0) ? 6 : (wave < 4) ? wave - 1 : wave - 2;      // worker number (wave 4: unused)
    const int slot = (wave < 4) ? wave : wave - 1;                           // panel-solve tile of this wave (wave 4 has none)
    if (!FIRST_DONE) {
        if (wave == 0) {
            const int bad = chol16_lp<Lay, false>(D, invd, lane, Xs, XS_LD);
            if (lane == 0 && bad && *s_bad == 0) *s_bad = bad;
        }
        __syncthreads();
    }
    d4 s0 = zero;                    // wave 0: its solved tile across the solve phase's barrier (the diagonal tile it updates
                                     // is NOT prefetched across it: eight more live registers slow chol16_lp by 0.3K cycles per tile)
    d4 keep[2] = {zero, zero};       // tiles of block row p-1 of X, carried into step p
    unsigned long long held = 0xFF;  // which ones (the two low fields of the plan word of the step that formed them)
    for (int p = 0; p <= npan; ++p) {
        const int c0 = p * 16;
        const double* Xp = Xs + (p & 1) * 16 * XS_LD;       // inverse of diagonal tile p
        const unsigned long long plan = (p < npan && wave != 4) ? c_fi_plan.w[npan][p][wid] : 0xFFull;
        FSTAMP(8 * p + 0);
        if (wave != 4) {
            // panel solve of tile (p + 1 + slot, p):  S = A X_p^T
            const int t = p + 1 + slot;
            if (p < npan && t < npan) {
                // Out[i][j] = sum_c X_p[rho(i)][c] A[j][c] = S[j][rho(i)]: the accumulator is fi_frag of the solved tile
                double* C = D + Lay::tile(t, p);
                const d4 a = fi_frag<Lay>(C, r, kq);
                const d4 x = fi_frag_xs(Xp, fi_rho(r), kq);
                d4 acc = zero;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s4], a[s4], acc, 0, 0, 0);
                if (wave != 0) {
                    fi_frag_store<Lay>(C, r, kq, acc);
#if !(FI_EXP & 4)
                    sink.tile(t, p, acc, lane);
#endif
                } else {
                    // acc[s] = S[r][kappa(s, kq)] is also the operand fragment s of the solved tile: wave 0 updates the tile its
                    // next 16x16 factorisation waits for from these registers -- AFTER the barrier that ends the solve phase
                    // (nobody else touches that tile), so that the other waves' update phase starts ~0.5K cycles earlier: the
                    // steps are bound by the workers, not by wave 0's chain
                    fi_frag_store<Lay>(C, r, kq, acc);
                    s0 = acc;
                }
            }
            // the tiles of block row p-1 of X held in registers since the previous step
#pragma unroll
            for (int cnt = 0; cnt < 2; ++cnt) {
                const int j = (int)(held >> (4 * cnt)) & 15;
                if (j != 15) fi_xstore<Lay>(D + Lay::tile(p - 1, j), keep[cnt], lane);
            }
        } else {
            if (p > 0) {
                const double* Xq = Xs + ((p - 1) & 1) * 16 * XS_LD;
                d4 xd;
#pragma unroll
                for (int g = 0; g < 4; ++g) xd[g] = Lay::XT ? Xq[r * XS_LD + kq + 4 * g] : Xq[(kq + 4 * g) * XS_LD + r];
                tile_write<Lay>(D + Lay::tile(p - 1, p - 1), xd, lane);
            }
        }
        FSTAMP(8 * p + 1);
        __syncthreads();
        FSTAMP(8 * p + 2);
        if (p == npan) break;
        const bool lastp = (p == npan - 1);
        held = 0xFF;
        if (wave == 4) {
            // block row p of L (final since the end of step p-1; the inverse overwrites it in the solve phase of step p+1)
            // goes back to HBM
#if !(FI_EXP & 4)
            sink.row(p, lane);
#endif
            WSTAMP(p, 0);
        } else if (wave != 0 || lastp) {
            // row p of the inverse: this worker's tiles
            held = plan & 0xFF;
#if !(FI_EXP & 2)
            if ((plan & 15) != 15) fi_inv_pair<Lay>(D, Xp, p, (int)plan & 15, (int)(plan >> 4) & 15, lane, keep[0], keep[1]);
#endif
            WSTAMP(p, 1);
            // its trailing tiles (rt >= ct > p), two at a time
#if FI_EXP & 8
            const int ntr = 0;
#else
            const int ntr = (int)(plan >> 8) & 15;
#endif
            for (int n = 0; n < ntr; ++n) {
                const int it = (int)(plan >> (12 + 12 * n)) & 0xFFF;
                const int rtA = it >> 9, ctA = (it >> 6) & 7, rtB = (it >> 3) & 7, ctB = it & 7;
                if (rtB == 0) fi_trail1<Lay>(D, p, rtA, ctA, lane);
                else if (rtA == rtB) fi_trail2<Lay, true>(D, p, rtA, ctA, rtB, ctB, lane);
                else fi_trail2<Lay, false>(D, p, rtA, ctA, rtB, ctB, lane);
            }
            WSTAMP(p, 2);
        }
        if (wave == 0 && !lastp) {
            double* Cd = D + Lay::tile(p + 1, p + 1);
            d4 c = tile_read<Lay>(Cd, lane);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-s0[s4], s0[s4], c, 0, 0, 0);
            tile_write<Lay>(Cd, c, lane);
        }
        if (wave == 0 && !lastp && !(FI_EXP & 1)) {
            FSTAMP(8 * p + 3);
            const int c1 = c0 + 16;
            const int bad = chol16_lp<Lay, false>(D + Lay::tile(p + 1, p + 1), invd + c1, lane, Xs + ((p + 1) & 1) * 16 * XS_LD, XS_LD);
            if (lane == 0 && bad && *s_bad == 0) *s_bad = c1 + bad;
            // wave 0 carries no tile of the inverse out of a step that has a 16x16 factorisation (it holds one only after
            // the LAST step): redefining `keep` here ends its live range at the top of this block, so that the register
            // allocator does not hold 16 registers across chol16_lp for the one wave that runs it
            keep[0] = zero;
            keep[1] = zero;
        }
        FSTAMP(8 * p + 4);
        __syncthreads();
        FSTAMP(8 * p + 5);
        WSTAMP(p, 3);
    }
}
