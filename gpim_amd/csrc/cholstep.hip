// cholstep.hip -- the blocked Cholesky as a single in-order stream of two kinds of launches (SURVEY 8(a) row a6;
// replaces torch.linalg.cholesky at gpim/gpreg/gpr.py:192,248).
//
// The diagonal-block factorisation (potf2, one workgroup, 128 serial pivots, ~36 us) is the one step of the
// factorisation that cannot be spread over the chip.  Instead of hiding it behind a second stream, its launch
// HOSTS the trailing update: workgroup 0 of `chol_step_kernel` factors block j, every other workgroup of the same
// launch computes one tile of pending trailing-update work with the MFMA tile engine ("fillers").  To make enough
// filler work independent of block j the update order is left-looking inside a window of two outer panels and
// right-looking beyond it:
//
//   H_j  (chol_step_kernel)   potf2(j)  ||  column j, rows i > j:  A[i,j] -= sum_{c in [p0(j)-W, j)} L[i,c] L[j,c]^T
//                                       ||  a share of bulk(P-1): tiles (i,jj), jj >= p0(P)+W, -= L[i,P-1] L[jj,P-1]^T
//   F_j  (panel_solve_kernel) L[i,j] = A[i,j] * inv(L[j,j])^T for all rows i > j, one-shot: the whole strip and the
//                             inverse (in MFMA operand order, written by potf2) are fetched with every load in
//                             flight at once -- no k-loop of load/barrier/compute rounds
//   D_j  (diag_update_kernel) the next diagonal tile:  A[j+1,j+1] -= L[j+1,j] L[j+1,j]^T  (its older window columns came
//                             with H_j's hosted tiles; the distributed panel chain and float handles update every
//                             diagonal tile of the window eagerly)
//
// P = outer panel of W = 4 block columns, p0/p1 its first / one-past-last column.  Per block column the chain is
// H_j -> F_j -> D_j; nothing else is ever on it.  Every tile receives each k-block exactly once: columns of the
// two most recent panels through the left-looking column updates (off-diagonal) or D (diagonal), older ones
// through bulk() -- see build_step_plan().
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include "potf2_body.hpp"
#include "gemm_body.hpp"

#ifndef STEP_W
#define STEP_W 4
#endif

// flag in the kb1 field of a hosted update tile (beside its kind, bits 16-18): store the result in StepArgs::Pcopy as well
#define TILE_COPY (1 << 19)

struct StepArgs {
    double* A; int64_t ld; int kblk; int nb;
    double* dinv_all; double* dinvB_all; double* logdet; int32_t* info;
    int col_off;                // global index of the sub-matrix's first column (non-PD reporting)
    int potf2;                  // 0: no factorisation role in this launch (hosted tiles only)
    // lock-step batches: the factorisation role works on problems pb_off .. pb_off + pb_cnt - 1, the hosted tiles on
    // problems hb_off .. hb_off + hb_cnt - 1 (two halves of a large batch take turns: launch_potrf_steps).  swap: the
    // problem index is blockIdx.x and the block index blockIdx.y -- workgroups are dispatched x-fastest, so every
    // problem's factorisation role starts before any hosted tile, and problem i stays on XCD i % 8.
    int pb_off, pb_cnt, hb_off, hb_cnt, swap;
    double* Tm;                 // != nullptr: fused inverse -- temporary of the T phases; the factorisation role writes
                                // the INVERSE of its block into A's diagonal block (a leaf of the inverse) instead of dinv_all
    double* Pcopy;              // B x 128 x 128: the update tile flagged TILE_COPY (tile (kblk+1, kblk) of the column update) is
                                // stored here too -- what panel_solve_diag_kernel reads instead of the block row it overwrites
    GemmArgs g;                 // hosted tiles; trailing-update tiles are NT with alpha = -1, beta = 1
};

// blockIdx.x == 0: potf2 of block kblk; 1..7: idle (keeps hosted workgroup b on XCD b % 8, which the tile engine's
// XCD-aware list mapping assumes); >= 8: hosted tile b - 8.  FTM x FTN: 128x128 = one workgroup per tile, 128x64 =
// one per row half, 64x64 = one per quadrant (few tiles: spread each over four CUs).
// A hosted tile's kind sits in bits 16.. of its kb1 field: 0 = trailing update A[ci,cj] -= L[ci,k] L[cj,k]^T; 1 / 2 =
// T phase of the triangular inverse, Tm[ci,cj] (=|+=) A[ci,k] A[k,cj]; 3 / 4 = X phase, A[ci,cj] (=|-=) -A[ci,k] Tm[k,cj]
// (plan_inverse below).  The factorisation role's 79 KB and 126 registers let two workgroups share a CU -- what the
// launches that host whole rounds of deep tiles want.  ALONE = the launch declares 84 KB instead, one workgroup per CU:
// where the chain of diagonal blocks bounds the launch, a hosted workgroup on the factorisation role's CU stretches
// the role from 35 to 50 us (its serial fp64 chain shares the SIMD's fp64 pipe with the neighbour's MFMAs), and what
// such a launch hosts for free is one quadrant per CU anyway.
template <int FTM, int FTN, bool ALONE, bool RAG = false>
__global__ __launch_bounds__(NTH, ALONE ? 2 : 4) void chol_step_kernel(StepArgs a) {
    constexpr int SM0 = POTF2_SMEM_DOUBLES > gemm_smem_doubles<FTM, FTN, 2>() ? POTF2_SMEM_DOUBLES : gemm_smem_doubles<FTM, FTN, 2>();
    static_assert(SM0 * 8 <= 80 * 1024, "two workgroups per CU");
    constexpr int SM = ALONE ? 84 * 128 : SM0;
    __shared__ __attribute__((aligned(16))) double smem[SM];
    const int b = a.swap ? blockIdx.y : blockIdx.x;
    const int by = a.swap ? blockIdx.x : blockIdx.y;
    if (b >= 8) {
        if (by >= a.hb_cnt) return;
        const int pb = by + a.hb_off;
        int quad;
        const int pos = gemm_tile_pos<FTM, FTN>(a.g.ntiles, a.g.chunk, b - 8, quad);
        TileDesc t = a.g.tiles[pos];
        const int kind = (t.kb1 >> 16) & 7;
        const bool copy = (t.kb1 & TILE_COPY) != 0;
        t.kb1 &= 0xffff;
        if (kind == 0) {
            if (copy && a.Pcopy) {
                GemmArgs g = a.g;
                g.C2 = a.Pcopy + (int64_t)pb * (NB * NB);
                gemm_tile_core<false, false, EPI_STORE, 8, FTM, FTN, 2, RAG>(g, t, quad, pb, smem);
            } else {
                gemm_tile_core<false, false, EPI_STORE, 8, FTM, FTN, 2, RAG>(a.g, t, quad, pb, smem);
            }
        } else {
            GemmArgs g = a.g;
            if (kind <= 2) { g.C = a.Tm; g.alpha = 1.0; g.beta = kind == 2 ? 1.0 : 0.0; }
            else { g.B = a.Tm; g.alpha = -1.0; g.beta = kind == 4 ? 1.0 : 0.0; }
            gemm_tile_core<false, true, EPI_STORE, 8, FTM, FTN, 2, RAG>(g, t, quad, pb, smem);
        }
        return;
    }
    if (b != 0 || !a.potf2 || by >= a.pb_cnt) return;
    // beside hosted neighbours on its CU the serial chain competes with their MFMA blocks (raised to priority 3 in the
    // tile engine): the role runs at that priority throughout (chain-bound launches: 1.93 -> 1.90 ms at N = 4212)
    if (!ALONE) __builtin_amdgcn_s_setprio(3);
    potf2_body<double, ALONE && FI_LOOKAHEAD>(smem, by + a.pb_off, a.A, a.ld, a.kblk, a.dinv_all, a.dinvB_all, a.logdet, a.info, a.nb,
                                              a.col_off, a.Tm != nullptr);
}

// Panel solve, one workgroup (4 waves) per 32-row strip of the block column below the diagonal block:
// S = P * Dinv^T, in place.  Wave w owns the 16-column tiles w and 7 - w of the strip (Dinv is lower triangular:
// tile t needs k-steps 0 .. 4t+3, so every wave runs 36 of them per 16 rows).
// (Strips of 16 and of 64 rows were measured in round 3: both slower than 32 -- potrf at N = 4212: 1.694 / 1.647 /
// 1.783 ms, at 16384: 29.65 / 28.96 / 29.60 -- the launch is bound by its load latency and by re-reading the inverse.)
// WPE = waves per SIMD the register allocation aims at: 2 for a single problem (the launch is latency-bound and well
// under one workgroup per CU), 4 for lock-step batches (> 1000 workgroups per launch: 129 registers allowed three per CU)
template <int WPE>
__global__ __launch_bounds__(256, WPE) void panel_solve_kernel(double* __restrict__ A, int64_t ld, int kblk, int nb,
                                                            const double* __restrict__ dinvB_all, int b_off) {
    constexpr int LDS_LD = 130;             // 130 % 32 == 2: the A-fragment reads below are bank-conflict free
    constexpr int ROWS = 32, MTS = ROWS / 16;
    __shared__ __attribute__((aligned(16))) double S[ROWS * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int by = blockIdx.y + b_off;          // problem of a lock-step batch
    A += (int64_t)by * nb * NB * ld;
    const d2* DB = reinterpret_cast<const d2*>(dinvB_all + ((int64_t)by * nb + kblk) * (NB * NB));
    double* P = A + ((int64_t)(kblk + 1) * NB + (int64_t)blockIdx.x * ROWS) * ld + (int64_t)kblk * NB;
    // the strip: one 1 KB row per wave-wide load, global -> LDS directly
#pragma unroll
    for (int i = 0; i < ROWS / 4; ++i) {
        const int row = wave * (ROWS / 4) + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P + (int64_t)row * ld + lane * 2),
                                         (__attribute__((address_space(3))) void*)(S + row * LDS_LD), 16, 0, 0);
    }
    // B fragments of both tiles: 18 sixteen-byte loads per lane, all in flight
    const int t0 = wave, t1 = 7 - wave, n0 = 2 * t0 + 2;
    d2 bf[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        const int t = q < n0 ? t0 : t1, s2 = q < n0 ? q : q - n0;
        bf[q] = DB[(t * 16 + s2) * 64 + lane];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
    d4 acc[2][MTS];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) acc[x][mt] = zero;
    const double* Sa = S + (lane & 15) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        const bool first = q < n0;          // wave-uniform
        const int s2 = first ? q : q - n0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int s = 2 * s2 + e;
#pragma unroll
            for (int mt = 0; mt < MTS; ++mt) {
                const double av = Sa[mt * 16 * LDS_LD + 4 * s];
                if (first) acc[0][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bf[q][e], acc[0][mt], 0, 0, 0);
                else acc[1][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bf[q][e], acc[1][mt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int t = x == 0 ? t0 : t1;
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                P[(int64_t)(mt * 16 + (lane >> 4) + 4 * rg) * ld + t * 16 + (lane & 15)] = acc[x][mt][rg];
    }
}

// Diagonal tiles (jj,jj), jj = kblk+1 .. kblk+ntile, -= L[jj,kblk] L[jj,kblk]^T, one workgroup (4 waves, one
// 16x16 MFMA tile each) per 32x32 quadrant of the lower half; one-shot like the panel solve: both 32x128
// operand strips go straight to LDS, the C values to registers, all loads in flight together.
__global__ __launch_bounds__(256, 2) void diag_update_kernel(double* __restrict__ A, int64_t ld, int kblk, int nb, int b_off) {
    constexpr int LDS_LD = 130;
    __shared__ __attribute__((aligned(16))) double S[64 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    A += (int64_t)(blockIdx.y + b_off) * nb * NB * ld;
    const int jj = kblk + 1 + blockIdx.x / 10, q = blockIdx.x % 10;
    // q -> (a, b), a >= b, a, b in 0..3
    const int a = q < 1 ? 0 : q < 3 ? 1 : q < 6 ? 2 : 3, b = q - a * (a + 1) / 2;
    const double* Pa = A + ((int64_t)jj * NB + a * 32) * ld + (int64_t)kblk * NB;
    const double* Pb = A + ((int64_t)jj * NB + b * 32) * ld + (int64_t)kblk * NB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 8 + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Pa + (int64_t)row * ld + lane * 2),
                                         (__attribute__((address_space(3))) void*)(S + row * LDS_LD), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Pb + (int64_t)row * ld + lane * 2),
                                         (__attribute__((address_space(3))) void*)(S + (32 + row) * LDS_LD), 16, 0, 0);
    }
    const int wm = wave >> 1, wn = wave & 1;
    double* C = A + ((int64_t)jj * NB + a * 32 + wm * 16 + (lane >> 4)) * ld + (int64_t)jj * NB + b * 32 + wn * 16 + (lane & 15);
    double cv[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) cv[rg] = C[(int64_t)(4 * rg) * ld];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;      // two chains: a lone dependent fp64 MFMA chain issues at half rate
    const double* Sa = S + (wm * 16 + (lane & 15)) * LDS_LD + (lane >> 4);
    const double* Sb = S + (32 + wn * 16 + (lane & 15)) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa[4 * s], Sb[4 * s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa[4 * s + 4], Sb[4 * s + 4], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) C[(int64_t)(4 * rg) * ld] = cv[rg] - (acc0[rg] + acc1[rg]);
}

// F_j and D_j in ONE launch (double precision, the next diagonal tile only).  512 threads: waves 0-3 and waves 4-7 each
// solve one ROWS-row strip (the arithmetic of panel_solve_kernel element by element, so the same bits).  The first NDIAG
// workgroups: the lower ROWS x ROWS blocks (a, b), a >= b, of the next diagonal block; each solves the strips a and b of
// block row kblk+1 ITSELF, keeps the results in LDS and applies its block of  A[jj,jj] -= S S^T  (one wave per 16x16 tile, the
// arithmetic of diag_update_kernel); block (a, a) also stores strip a of L.  The other workgroups: two strips each of the
// block rows below.  The strips of block row kblk+1 are read from Pcopy -- the copy the step launch made of A[kblk+1, kblk]
// -- because several workgroups read a strip that one of them overwrites.
//   ROWS = 16 (36 blocks): every workgroup has the MFMA work of ONE 32-row strip of panel_solve_kernel; the redundant solves run
//              on CUs that a chain-bound launch leaves idle: 12.1 us + a boundary -> 8 us per block column (N = 4212: 2.41 ->
//              2.33 ms per Adam iteration);
//   ROWS = 32 (10 blocks): half the workgroups, for the launches with many block rows below (one round of the chip at
//              N = 16384), where what counts is that the chip no longer idles through D_j: 72.2 -> 71.85 ms.
template <int ROWS>
__global__ __launch_bounds__(512, 2) void panel_solve_diag_kernel(double* __restrict__ A, int64_t ld, int kblk, int nb,
                                                                  const double* __restrict__ dinvB_all,
                                                                  const double* __restrict__ Pcopy, int b_off) {
    constexpr int LDS_LD = 130, MTS = ROWS / 16, NS = NB / ROWS, NDIAG = NS * (NS + 1) / 2;
    __shared__ __attribute__((aligned(16))) double S[2 * ROWS * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, half = tid >> 8;
    const int by = blockIdx.y + b_off;
    A += (int64_t)by * nb * NB * ld;
    Pcopy += (int64_t)by * (NB * NB);
    const d2* DB = reinterpret_cast<const d2*>(dinvB_all + ((int64_t)by * nb + kblk) * (NB * NB));
    const bool diag = blockIdx.x < NDIAG;
    int a = 0, b = 0;
    if (diag) {
        const int q = blockIdx.x;
        while ((a + 1) * (a + 2) / 2 <= q) ++a;
        b = q - a * (a + 1) / 2;
    }
    const int jj = kblk + 1;
    // this half's strip: strip a / b of block row jj, or one of the strips of the rows below
    const int strip = diag ? (half == 0 ? a : b) : NS + 2 * (int)(blockIdx.x - NDIAG) + half;
    const bool work = !diag || half == 0 || b != a;           // (block (a, a): the second half has nothing to solve)
    double* P = A + ((int64_t)jj * NB + strip * ROWS) * ld + (int64_t)kblk * NB;
    double* Ss = S + half * ROWS * LDS_LD;
    // the strip: one 1 KB row per wave-wide load, global -> LDS directly
    if (work) {
#pragma unroll
        for (int i = 0; i < ROWS / 4; ++i) {
            const int row = wave * (ROWS / 4) + i;
            const double* src = diag ? Pcopy + (strip * ROWS + row) * NB + lane * 2 : P + (int64_t)row * ld + lane * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Ss + row * LDS_LD), 16, 0, 0);
        }
    }
    // B fragments of both column tiles of this wave (panel_solve_kernel)
    const int t0 = wave, t1 = 7 - wave, n0 = 2 * t0 + 2;
    d2 bf[18];
    if (work) {
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const int t = q < n0 ? t0 : t1, s2 = q < n0 ? q : q - n0;
            bf[q] = DB[(t * 16 + s2) * 64 + lane];
        }
    }
    // the 16x16 tile of the diagonal block this wave updates (the first MTS x MTS waves of the workgroup)
    const int wt = tid >> 6, wm = wt / MTS, wn = wt % MTS;
    const bool syrk = diag && wt < MTS * MTS;
    double* C = A + ((int64_t)jj * NB + a * ROWS + wm * 16 + (lane >> 4)) * ld + (int64_t)jj * NB + b * ROWS + wn * 16 + (lane & 15);
    double cv[4] = {0.0, 0.0, 0.0, 0.0};
    if (syrk) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) cv[rg] = C[(int64_t)(4 * rg) * ld];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
    d4 acc[2][MTS];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) acc[x][mt] = zero;
    if (work) {
        const double* Sa = Ss + (lane & 15) * LDS_LD + (lane >> 4);
#pragma unroll
        for (int q = 0; q < 18; ++q) {
            const bool first = q < n0;          // wave-uniform
            const int s2 = first ? q : q - n0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = 2 * s2 + e;
#pragma unroll
                for (int mt = 0; mt < MTS; ++mt) {
                    const double av = Sa[mt * 16 * LDS_LD + 4 * s];
                    if (first) acc[0][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bf[q][e], acc[0][mt], 0, 0, 0);
                    else acc[1][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bf[q][e], acc[1][mt], 0, 0, 0);
                }
            }
        }
        // the solved strip: to L (the strips below, and strip a from block (a, a)) ...
        if (!diag || (a == b && half == 0)) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int t = x == 0 ? t0 : t1;
#pragma unroll
                for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        P[(int64_t)(mt * 16 + (lane >> 4) + 4 * rg) * ld + t * 16 + (lane & 15)] = acc[x][mt][rg];
            }
        }
    }
    if (!diag) return;                          // (the whole workgroup)
    // ... and over the strip it came from in LDS (every wave of the half has read all of it: barrier first)
    __syncthreads();
    if (work) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int t = x == 0 ? t0 : t1;
#pragma unroll
            for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    Ss[(mt * 16 + (lane >> 4) + 4 * rg) * LDS_LD + t * 16 + (lane & 15)] = acc[x][mt][rg];
        }
    }
    __syncthreads();
    if (!syrk) return;
    d4 acc0 = zero, acc1 = zero;                // two chains, as diag_update_kernel
    const double* Sa = S + (wm * 16 + (lane & 15)) * LDS_LD + (lane >> 4);
    const double* Sb = S + ((b != a ? ROWS : 0) + wn * 16 + (lane & 15)) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa[4 * s], Sb[4 * s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa[4 * s + 4], Sb[4 * s + 4], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) C[(int64_t)(4 * rg) * ld] = cv[rg] - (acc0[rg] + acc1[rg]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static void lower_patches(std::vector<TileDesc>& out, int lo, int hi, int kb0, int kb1) {
    for (int ig = lo / 8; ig <= (hi - 1) / 8; ++ig)
        for (int jg = lo / 8; jg <= ig; ++jg)
            for (int i = std::max(lo, ig * 8); i < std::min(hi, ig * 8 + 8); ++i)
                for (int j = std::max(lo, jg * 8); j < std::min(hi, jg * 8 + 8); ++j)
                    if (j <= i) out.push_back({i, j, kb0, kb1});
}

// Hosting policy of single-precision handles (cholstep32.hip; tools/r3_f32_sweep2.sh): a hosted float tile takes half
// the matrix-core time beside a factorisation role of unchanged length.  Everything is hosted up to nb = 87, 128 tiles
// per launch beyond (64 in pair mode), the rest of a panel's bulk update runs as a tile-engine launch before the
// panel's first step; pair mode (panels applied two at a time, k-depth 1024) from nb = 160.
static bool pair_mode32(int nb) { return nb >= 160; }
static int fill_cap32(int nb) { return pair_mode32(nb) ? 64 : (nb < 88 ? (1 << 30) : 128); }

// List scheduling on the chip's workgroup slots, the order the hardware dispatches the workgroups of a launch in
// (512 slots: two 8-wave workgroups per CU).  Costs in units of one 128-deep k-block of a 128x128 tile per slot
// (~34 us at two workgroups per CU; the factorisation role lasts ~1.1).
struct HostSim {
    std::vector<double> slot;       // min-heap of slot finish times
    double makespan = 0.0;
    explicit HostSim(int S) : slot(S, 0.0) {}
    static double cost(const TileDesc& t) { return ((t.kb1 & 0xffff) - t.kb0) + 0.5; }     // k-blocks + prologue / epilogue
    double peek() const { return slot.front(); }
    void add(double c) {
        std::pop_heap(slot.begin(), slot.end(), std::greater<double>());
        slot.back() += c;
        makespan = std::max(makespan, slot.back());
        std::push_heap(slot.begin(), slot.end(), std::greater<double>());
    }
    // q workgroups of cost c each, on the q earliest slots, only if ALL of them end by `limit` (a tile operation is q
    // workgroups: testing the earliest slot alone let the last one or two spill into a second round of the chip, +10-14 us
    // on a 36 us launch)
    bool try_add(double c, int q, double limit) {
        double t[4];
        for (int w = 0; w < q; ++w) {
            std::pop_heap(slot.begin(), slot.end() - w, std::greater<double>());
            t[w] = slot[slot.size() - 1 - w];
        }
        const bool ok = t[q - 1] + c <= limit;       // t[] ascends: the q-th earliest slot is the latest of them
        for (int w = q - 1; w >= 0; --w) {
            if (ok) { slot[slot.size() - 1 - w] = t[w] + c; makespan = std::max(makespan, t[w] + c); }
            std::push_heap(slot.begin(), slot.end() - w, std::greater<double>());
        }
        return ok;
    }
};
#define HOST_SLOTS 512
// one hosted workgroup of a tile operation `depth` k-blocks deep in shape q (4: 64x64 quadrant, 2: 128x64 half, 1: whole
// tile), from kernel traces at two workgroups per CU: quadrants 8 + 6 depth us, whole tiles 34 us per k-block
// (quadrants re-measured in round 4 on launches whose lists are dispatched deepest-first: depth 7 lasts 41 us, depth 4
// 27 -- 0.14 units per k-block, not 0.18; with it and chunks of up to 6 k-blocks N = 4212 runs 2.63 -> 2.59 ms per Adam
// iteration, N = 8192 11.75 -> 11.59; 0.13 / 0.15, chunks of 5 / 7 and a budget of 1.05 / 1.2 units are all slower)
static double wg_cost(int depth, int q) { return q == 4 ? 0.14 * depth + 0.24 : (q == 2 ? 0.34 * depth + 0.3 : depth + 0.5); }

// ---- what the step launches host of the TRAILING UPDATE (double precision) -------------------------------------------
// nb < 64: the chain of diagonal blocks bounds the factorisation; every launch hosts its column update (left-looking
// inside a window of two panels, k-depth <= 7) and a quarter of the previous panel's bulk update (k-depth 4) -- the
// round-3 lists, unchanged (same bits).
// nb >= 64: everything is hosted as well (two workgroups per CU since the factorisation role fits 79 KB), and what a
// launch hosts is chosen so that it ends on a full round of the 512 slots:
//   * tile (i, jj) keeps the first block column it has not yet received (`pend`); a flush applies [pend, p0) in ONE
//     tile operation, so a tile that is skipped for a panel comes back twice as deep (k = 1024: half the
//     read-modify-write passes over C, half the prologues -- what round 3's "pair mode" did for whole panels);
//   * tiles of the NEXT panel's columns must be flushed during this panel (deadline), the others are optional;
//   * steady state: the two patch classes ((i / 8 + jj / 8) & 1) take turns, each flushed every other panel at depth
//     8 -- every panel carries the same amount of work -- and the holes of a launch's last round are filled with
//     whatever else is pending;
//   * once fewer than 64 block columns remain a panel's updates are less than one round per launch and everything
//     pending is flushed at once, evenly over the panel's launches.
static void plan_updates(int nb, std::vector<std::vector<TileDesc>>& fill) {
    // chain-bound matrices of at least 16 block columns: panels of THREE (column updates <= 5 k-blocks deep: a hosted quadrant
    // of depth 7 lasts 50 us, the factorisation role 28 since round 5 -- N = 4212 2.53 -> 2.47 ms per Adam iteration; the
    // ten-block slices of config C3 in one lock-step batch of 64 are 1.4 % slower with it, hence the lower bound)
#ifndef STEP_W_CHAIN
#define STEP_W_CHAIN 3
#endif
#ifndef STEP_W_CHAIN_MIN
#define STEP_W_CHAIN_MIN 16
#endif
    const int W = (nb >= STEP_W_CHAIN_MIN && nb < 64) ? STEP_W_CHAIN : STEP_W, S = HOST_SLOTS;
    const int npanel = (nb + W - 1) / W;
    auto by_depth = [](const TileDesc& a, const TileDesc& b) { return (a.kb1 - a.kb0) > (b.kb1 - b.kb0); };
    // left-looking column update of block column j inside a window of two outer panels: ONE tile operation per tile,
    // k-blocks [p0 - W, j).  (Round 4 also measured it split at the panel boundary -- every hosted tile <= 4 k-blocks
    // deep, the previous panel's part one launch earlier: launches that host nothing else drop from 50 to 35 us, but
    // the two passes over C and the smaller budget for the inverse's chunks give it back: N = 4212 835 vs 829 ms per
    // 300 iterations, lock-step batches of config C3 3 % slower.)
    // The NEXT diagonal tile (j+1, j+1) rides along: its window columns [kb0', j) are final when H_j starts; D_j adds
    // column j, the only one on the critical path.  (Rounds 3-4 had D_j update the next <= 7 diagonal tiles eagerly,
    // k = 128 each: the same flop as 10 quadrant workgroups per tile and step -- 2240 latency-shaped workgroups per launch
    // for a lock-step batch of 32, 28 us where the chip does 0.6 GFLOP.)
    auto col_update = [&](std::vector<TileDesc>& tl, int p0, int j) {
        const int kb0 = std::max(0, p0 - W);
        if (j > kb0)
            for (int i = j + 1; i < nb; ++i) tl.push_back({i, j, kb0, j});
        if (j + 1 < nb) {
            const int kd = std::max(0, ((j + 1) / W) * W - W);
            if (j > kd) tl.push_back({j + 1, j + 1, kd, j});
        }
    };
    if (nb < 64) {
        for (int p = 0; p < npanel; ++p) {
            const int p0 = p * W, p1 = std::min(p0 + W, nb), ncol = p1 - p0;
            std::vector<TileDesc> bulk;
            if (p > 0 && p0 + W < nb) lower_patches(bulk, p0 + W, nb, p0 - W, p0);
            const size_t per = (bulk.size() + ncol - 1) / ncol;
            size_t taken = 0;
            for (int j = p0; j < p1; ++j) {
                col_update(fill[j], p0, j);
                for (size_t q = 0; q < per && taken < bulk.size(); ++q) fill[j].push_back(bulk[taken++]);
            }
        }
        return;
    }
    std::vector<int> pend((size_t)nb * nb, 0);        // first source block column tile (i, jj) has not received via bulk
    for (int p = 0; p < npanel; ++p) {
        const int p0 = p * W, p1 = std::min(p0 + W, nb);
        const int rem = nb - p0 - W;
        std::vector<TileDesc> req, opt;
        if (p > 0 && rem > 0) {
            const int c0 = p0 + W, c1 = std::min(p0 + 2 * W, nb);
            for (int i = c0; i < nb; ++i)
                for (int jj = c0; jj < std::min(c1, i + 1); ++jj) req.push_back({i, jj, pend[(size_t)i * nb + jj], p0});
            if (c1 < nb) lower_patches(opt, c1, nb, 0, p0);
            for (auto& t : opt) t.kb0 = pend[(size_t)t.ci * nb + t.cj];
        }
        const bool bulk_regime = rem >= 64;
#ifndef PLAN_FLUSH
#define PLAN_FLUSH 2
#endif
        // patch classes: a tile is flushed every F-th panel, F * W k-blocks deep.  (Round 5 measured F = 3 / 4 -- depth 12 / 16,
        // fewer passes over C -- at N = 16384: 72.6 / 73.7 ms per iteration against 72.1, N = 12288 33.0 / 32.85 against 32.8:
        // the rounds of a launch get longer and its last one emptier)
        constexpr int F = PLAN_FLUSH;
        auto due = [&](const TileDesc& t) { return (t.kb1 - t.kb0) >= F * W || (((t.ci >> 3) + (t.cj >> 3)) % F) == (p % F); };
        double due_cost = 0.0;                 // cost of the due tiles not yet hosted
        size_t n_due = 0;
        if (bulk_regime) {
            // due first (deepest first), then the rest; each group in patch order
            std::stable_sort(opt.begin(), opt.end(), [&](const TileDesc& a, const TileDesc& b) {
                const int da = due(a), db = due(b);
                if (da != db) return da > db;
                return da ? by_depth(a, b) : false;
            });
            for (auto& t : opt)
                if (due(t)) { ++n_due; due_cost += HostSim::cost(t); }
        }
        size_t rtaken = 0, otaken = 0;
        for (int j = p0; j < p1; ++j) {
            const int left = p1 - j;
            std::vector<TileDesc>& tl = fill[j];
            col_update(tl, p0, j);
            const size_t nreq = (req.size() - rtaken + left - 1) / left;
            for (size_t q = 0; q < nreq; ++q) tl.push_back(req[rtaken++]);
            if (!bulk_regime) {
                const size_t nopt = (opt.size() - otaken + left - 1) / left;
                for (size_t q = 0; q < nopt; ++q) tl.push_back(opt[otaken++]);
            } else {
                HostSim sim(S);
                sim.add(1.0);                                               // the factorisation role
                std::stable_sort(tl.begin(), tl.end(), by_depth);
                double base = 0.0;
                for (auto& t : tl) { sim.add(HostSim::cost(t)); base += HostSim::cost(t); }
                // This launch's length: its share of the due work, rounded to whole deep tiles per slot (a slot runs
                // a sequence of tiles; 8.5 = one tile of depth 8) and never shorter than what it must host anyway.
                const double per_slot = (base + due_cost / left) / S;
                const double unit = F * W + 0.5;
                const double target = std::max(sim.makespan, unit * std::max(1.0, std::floor(per_slot / unit + 0.5)));
                double got = 0.0;
                // due tiles first (the sorted list starts with them), then anything else that still ends in time
                while (otaken < opt.size() && sim.peek() + HostSim::cost(opt[otaken]) <= target + 0.25) {
                    const double c = HostSim::cost(opt[otaken]);
                    sim.add(c);
                    if (otaken < n_due) got += c;
                    tl.push_back(opt[otaken++]);
                }
                due_cost -= std::min(due_cost, got);
            }
            // dispatch order = list order: deepest tiles first
            std::stable_sort(tl.begin(), tl.end(), by_depth);
            for (auto& t : tl)
                if (t.cj >= p0 + W) pend[(size_t)t.ci * nb + t.cj] = t.kb1;
        }
    }
}

// ---- the triangular inverse as hosted work -----------------------------------------------------------------------------
// X = L^-1 by recursive halving: node [lo, mid, hi) needs  T = L21 X11  (T phase; Tm[ci, cj] = sum_{k in [cj, mid)}
// L[ci, k] X[k, cj], ci in [mid, hi), cj in [lo, mid)) and then  X21 = -X22 T  (X phase; A[ci, cj] = -sum_{k in [mid, ci]}
// X[ci, k] Tm[k, cj], in place of L21).  The leaves are the inverses of the diagonal blocks, which the factorisation
// role writes into A's diagonal blocks itself.  Instead of 2 log2(nb) launches AFTER the factorisation, the tile
// operations of a node are handed to the step launches as soon as their operands are final:
//   T phase of a node: block columns < mid factored and solved (launch index >= mid) and the left child inverted;
//   X phase: T phase complete, the right child inverted, block rows < hi dead for the factorisation (index >= hi).
// A step launch that the chain of diagonal blocks bounds (most of them at mid-size N, the last third at N = 16384)
// has idle workgroup slots for about the length of the factorisation role: it hosts chunks of <= 6 k-blocks (one chunk
// of a tile per launch, accumulated into the output across launches) up to that length.  What is left when the
// factorisation ends -- the X phases of the nodes that contain the last block column, and most of the root's work at
// large N -- runs as a few launches of the same kernel without the factorisation role.
enum { TK_UPDATE = 0, TK_T_FIRST = 1, TK_T_ACC = 2, TK_X_FIRST = 3, TK_X_ACC = 4 };
struct TriOp { int ci, cj, k0, k1, kp; };
struct TriNodeS {
    int lo, mid, hi, left, right, height;
    int t_done = 1 << 30, x_done = 1 << 30;     // launch index that issues the last chunk of the phase
    std::vector<TriOp> T, X;
    size_t t_open = 0, x_open = 0;
};
static int tri_nodes(int lo, int hi, std::vector<TriNodeS>& nodes) {
    TriNodeS n;
    n.lo = lo; n.hi = hi; n.mid = lo; n.left = n.right = -1; n.height = 0;
    if (hi - lo <= 1) {
        n.t_done = n.x_done = lo;               // the factorisation role of launch `lo` writes the block's inverse
        nodes.push_back(n);
        return (int)nodes.size() - 1;
    }
    n.mid = lo + (hi - lo + 1) / 2;      // (left-heavy splits, mid = lo + 0.6 ... 0.85 of the node, leave MORE of the inverse after the last step: 3.2K -> 3.2K / 3.6K / 4.6K / 6.1K k-blocks at nb = 33)
    n.left = tri_nodes(lo, n.mid, nodes);
    n.right = tri_nodes(n.mid, hi, nodes);
    n.height = 1 + std::max(nodes[n.left].height, nodes[n.right].height);
    for (int cj = lo; cj < n.mid; ++cj)         // longest k-ranges first
        for (int ci = n.mid; ci < hi; ++ci) n.T.push_back({ci, cj, cj, n.mid, cj});
    for (int ci = hi - 1; ci >= n.mid; --ci)
        for (int cj = lo; cj < n.mid; ++cj) n.X.push_back({ci, cj, n.mid, ci + 1, n.mid});
    n.t_open = n.T.size();
    n.x_open = n.X.size();
    nodes.push_back(n);
    return (int)nodes.size() - 1;
}

// shape of a step launch's hosted workgroups by the k-blocks of trailing update it hosts (times the batch):
// 4 = 64x64 quadrants (the work of a chain-bound launch spread over the chip), 2 = 128x64 halves, 1 = whole tiles.
// (Round 3 drew the lines at 128 / 512 tiles of average depth 5; tools/r3_exp*.sh.)
// (Round 4 re-measured the first line with deepest-first lists: 2000 instead of 700 -- N = 6000 5.50 -> 5.43 ms per Adam
// iteration, 8192 11.57 -> 11.43, 12288 33.27 -> 32.97, 16384 72.56 -> 72.28; 1100 / 1500 / 2400 / 2800 lie between.)
#ifndef BATCH_QUAD_MAX
#define BATCH_QUAD_MAX 700
#endif
static int host_shape(int64_t update_kblocks, int quad_max = 2000) { return update_kblocks <= quad_max ? 4 : (update_kblocks <= 2800 ? 2 : 1); }

// pair[L] (out): the step launch L runs its hosted quadrants two per CU (launch_step) -- decided here from the update
// tiles alone, so that the budget for the inverse's chunks is the one of the launch as it will run.
// (>= 300 update quadrants of average depth >= 3; 420 / 3.5 before the planner knew: N = 8192 11.42 -> 11.31 ms per
// iteration; 240 / 2.5, depth 2 and never pairing -- 12.15 ms -- are slower)
#ifndef PAIR_MIN_QUADS
#define PAIR_MIN_QUADS 300
#endif
#ifndef PAIR_MIN_DEPTH10
#define PAIR_MIN_DEPTH10 30
#endif
static bool pair_rule(size_t ntiles, int64_t kblocks) { return 4 * ntiles >= PAIR_MIN_QUADS && 10 * kblocks >= PAIR_MIN_DEPTH10 * (int64_t)ntiles; }
// returns false if the plan did not close (every launch completes at least one phase of one node, so this cannot happen
// with the dependency rule as it stands; a caller must not run a partial plan: A would be left a half-inverted factor)
static bool plan_inverse(int nb, std::vector<std::vector<TileDesc>>& fill, std::vector<std::vector<TileDesc>>& post,
                         std::vector<uint8_t>& pair) {
    std::vector<TriNodeS> nodes;
    tri_nodes(0, nb, nodes);
    std::vector<int> order(nodes.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return nodes[a].height < nodes[b].height; });
    size_t open_nodes = 0;
    for (auto& n : nodes) open_nodes += (n.height > 0);
#ifndef PLAN_CHAIN
#define PLAN_CHAIN 0.8
#endif
    const double CHAIN = PLAN_CHAIN;                   // length of the factorisation role in cost units
    for (int L = 0; open_nodes > 0; ++L) {
        const bool hosted = L < nb;
        if (!hosted) post.emplace_back();
        std::vector<TileDesc>& out = hosted ? fill[L] : post.back();
        int64_t upd = 0;
        for (auto& t : out) upd += t.kb1 - t.kb0;
        const int q = hosted ? host_shape(upd) : 1;
        const int cd = !hosted ? (1 << 20) : 6;
        // quadrants / halves: what a chain-bound launch hosts for free is about ONE workgroup per CU (traces at
        // N = 4212: up to ~250 quadrant workgroups leave the launch at the factorisation role's 35 us, 400 make it 50)
        // ... unless the update tiles alone need two rounds of deep quadrants: that launch runs two workgroups per CU
        // (twice the slots, every workgroup ~1.7x as long, the factorisation role stretched to ~48 us)
        // (N = 8192: 11.48 -> 11.37 ms per iteration with the budget of those launches modelled this way; the two
        // constants matter little: 1.5 ... 2.0 and 1.3 ... 1.6 measure the same)
        const bool two = hosted && q == 4 && pair_rule(out.size(), upd);
        if (hosted) pair[L] = two;
#ifndef PLAN_PAIR_SLOW
#define PLAN_PAIR_SLOW 1.7
#endif
#ifndef PLAN_PAIR_CHAIN
#define PLAN_PAIR_CHAIN 0.3
#endif
        const double slow = two ? PLAN_PAIR_SLOW : 1.0, chain = two ? CHAIN + PLAN_PAIR_CHAIN : CHAIN;
        HostSim sim(q == 1 || two ? HOST_SLOTS : HOST_SLOTS / 2);
        double target = 1e30;
        if (hosted) {
            sim.add(chain);
            for (auto& t : out)
                for (int w = 0; w < q; ++w) sim.add(slow * wg_cost(t.kb1 - t.kb0, q));
            target = std::max(sim.makespan, chain);
            // (round 5: a larger budget for the inverse's chunks in the second half of a large factorisation -- + 0.5 / 1 / 2
            // units from launch nb / 2 or 3 nb / 4 on -- measured 72.5 - 72.8 ms per iteration at N = 16384 against 72.25:
            // what a launch hosts beyond its own length runs at the hosted tiles' 60 TFLOP/s, after the last step at 69)
        }
        bool full = false;
        for (int idx : order) {
            TriNodeS& n = nodes[idx];
            if (n.height == 0 || n.x_open == 0 || full) continue;
            const bool t_phase = n.t_open > 0;
            if (t_phase ? !(nodes[n.left].x_done < L && n.mid <= L) : !(n.t_done < L && nodes[n.right].x_done < L && n.hi <= L)) continue;
            std::vector<TriOp>& ops = t_phase ? n.T : n.X;
            size_t& open = t_phase ? n.t_open : n.x_open;
            for (auto& o : ops) {
                if (o.kp >= o.k1) continue;
                int depth = std::min(o.k1 - o.kp, cd);
                if (hosted) {
                    // the deepest chunk ALL of whose workgroups still end with the launch
                    while (depth >= 1 && !sim.try_add(slow * wg_cost(depth, q), q, target + 0.02)) --depth;
                    if (depth < 1) { full = true; break; }
                } else {
                    for (int w = 0; w < q; ++w) sim.add(wg_cost(depth, q));
                }
                const int k1 = o.kp + depth;
                const int kind = t_phase ? (o.kp == o.k0 ? TK_T_FIRST : TK_T_ACC) : (o.kp == o.k0 ? TK_X_FIRST : TK_X_ACC);
                out.push_back({o.ci, o.cj, o.kp, k1 | (kind << 16)});
                o.kp = k1;
                if (o.kp >= o.k1 && --open == 0) {
                    if (t_phase) n.t_done = L;
                    else { n.x_done = L; --open_nodes; }
                }
            }
        }
        // dispatch order = list order: the deepest workgroups first.  A launch of more workgroups than the chip has slots
        // packs two shallow chunks into a slot -- as the simulation above assumed -- only if the deep ones are not left
        // for the end (chunks of the last node handled used to be: launches of 260-290 quadrants lasted 49 us, not 36-39)
        std::stable_sort(out.begin(), out.end(), [](const TileDesc& a, const TileDesc& b) {
            return ((a.kb1 & 0xffff) - a.kb0) > ((b.kb1 & 0xffff) - b.kb0);
        });
        if (!hosted && out.empty()) post.pop_back();      // (dependencies only resolve across launches)
        if (L > nb + 4 * 64) return false;
    }
    return true;
}

static int step_plan_build(gpimhip_ctx* h, int nb, StepPlan& P, bool with_inverse) {
    const bool fp32 = h->fp32 != 0;
    if (P.nb == nb && P.fp32 == (int)fp32) return GPIMHIP_OK;
    if (P.d_tiles) { (void)hipFree(P.d_tiles); P.d_tiles = nullptr; }
    const int W = STEP_W;
    const int npanel = (nb + W - 1) / W;
    std::vector<std::vector<TileDesc>> fill(nb), post, rest(npanel);
    if (!fp32) {
        plan_updates(nb, fill);
    } else {
        for (int p = 0; p < npanel; ++p) {
            const int p0 = p * W, p1 = std::min(p0 + W, nb), ncol = p1 - p0;
            // bulk(p-1): columns >= p0 + W, k-blocks = the columns of panel p-1.
            // Pair mode (large matrices): panels are applied two at a time to everything at least two panels to their
            // right -- k-depth 1024, half the passes over the trailing matrix -- and an even panel alone only to the
            // one destination panel that needs it before its partner is factored:
            //   start of an odd panel p  (p-1 even): panel p-1 -> destination panel p+1 only           (k-depth 512)
            //   start of an even panel p (p-1 odd) : panels p-2, p-1 -> every column >= p0 + W          (k-depth 1024)
            std::vector<TileDesc> bulk;
            if (p > 0 && p0 + W < nb) {
                if (!pair_mode32(nb)) {
                    lower_patches(bulk, p0 + W, nb, p0 - W, p0);
                } else if ((p - 1) % 2 == 0) {
                    const int c0 = p0 + W, c1 = std::min(p0 + 2 * W, nb);
                    for (int ig = c0 / 8; ig <= (nb - 1) / 8; ++ig)
                        for (int i = std::max(c0, ig * 8); i < std::min(nb, ig * 8 + 8); ++i)
                            for (int j = c0; j < std::min(c1, i + 1); ++j) bulk.push_back({i, j, p0 - W, p0});
                } else {
                    lower_patches(bulk, p0 + W, nb, p0 - 2 * W, p0);
                }
            }
            const size_t per = std::min<size_t>((size_t)fill_cap32(nb), (bulk.size() + ncol - 1) / ncol);
            size_t taken = 0;
            for (int j = p0; j < p1; ++j) {
                const int kb0 = std::max(0, p0 - W);
                if (j > kb0)
                    for (int i = j + 1; i < nb; ++i) fill[j].push_back({i, j, kb0, j});
                for (size_t q = 0; q < per && taken < bulk.size(); ++q) fill[j].push_back(bulk[taken++]);
            }
            while (taken < bulk.size()) rest[p].push_back(bulk[taken++]);
        }
    }
    P.n_update.assign(nb, 0);
    for (int j = 0; j < nb; ++j)
        for (auto& t : fill[j]) P.n_update[j] += t.kb1 - t.kb0;
    P.pair.assign(nb, 0);
    if (with_inverse) {
        // plan_inverse decides pair[] from the update tiles alone (before it adds its own) and budgets its chunks for
        // the launch as it will then run: keep ITS decision
        if (!plan_inverse(nb, fill, post, P.pair)) {
            gpim_set_error("step plan: the inverse's tile operations did not all find a launch (nb = " + std::to_string(nb) + ")");
            return GPIMHIP_E_BADARG;
        }
        // (rounds 4-5 overwrote it here by a dangling else with the rule below applied to the lists INCLUDING the inverse's
        // tiles; A/B on one box, round 6: N = 4212 2.337 -> 2.302 ms per Adam iteration, 8192 11.06 -> 10.89, 16384 71.86 -> 71.64)
    } else {
        for (int j = 0; j < nb; ++j) P.pair[j] = host_shape(P.n_update[j]) == 4 && pair_rule(fill[j].size(), P.n_update[j]);
    }
    P.n_all.assign(nb, 0);
    for (int j = 0; j < nb; ++j)
        for (auto& t : fill[j]) P.n_all[j] += (t.kb1 & 0xffff) - t.kb0;
    std::vector<TileDesc> tl;
    auto put = [&](const std::vector<TileDesc>& v) {
        PlanRange r{(int64_t)tl.size(), (int32_t)v.size()};
        tl.insert(tl.end(), v.begin(), v.end());
        return r;
    };
    P.fill.assign(nb, {0, 0});
    P.diag.assign(nb, {0, 0});
    P.bulk_rest.assign(npanel, {0, 0});
    P.post.clear();
    P.copy.assign(nb, 0);
    for (int j = 0; j < nb; ++j) {
        // the column update's tile (j+1, j) -- the last operation on it, in the step launch of column j (both halves of a
        // split batch host it before their F_j) -- also goes to the copy that panel_solve_diag_kernel reads
        if (!fp32)
            for (TileDesc& t : fill[j])
                if ((t.kb1 >> 16) == TK_UPDATE && t.ci == j + 1 && t.cj == j) { t.kb1 |= TILE_COPY; P.copy[j] = 1; }
        P.fill[j] = put(fill[j]);
        const int p1 = std::min((j / W) * W + W, nb);
        // diagonal tiles D_j updates (count): double precision the next one only (the others receive column j through
        // the hosted window update of their own step, plan_updates); float handles every tile of the window, eagerly
        P.diag[j] = PlanRange{0, fp32 ? std::max(0, std::min(nb, p1 + W) - (j + 1)) : (j + 1 < nb ? 1 : 0)};
    }
    for (int p = 0; p < npanel; ++p) P.bulk_rest[p] = put(rest[p]);
    for (auto& v : post) P.post.push_back(put(v));
    P.n_tiles = (int64_t)tl.size();
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(tl.size(), 1) * sizeof(TileDesc)));
    P.d_tiles = (TileDesc*)q;
    HIP_TRY(hipMemcpyAsync(P.d_tiles, tl.data(), tl.size() * sizeof(TileDesc), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    P.nb = nb;
    P.fp32 = (int)fp32;
    return GPIMHIP_OK;
}

// The double-precision plan as host data, for tests (no GPU involved): records of six int32 -- launch index (< nb: the
// step launch of that block column; >= nb: the launches after the last step), ci, cj, kb0, kb1, kind.
extern "C" int gpimhip_step_plan_host(int32_t nb, int32_t with_inverse, int32_t* out, int64_t cap, int64_t* n_out) {
    if (nb < 1 || nb > 4096 || !n_out) return GPIMHIP_E_BADARG;
    std::vector<std::vector<TileDesc>> fill(nb), post;
    plan_updates(nb, fill);
    std::vector<uint8_t> pair(nb, 0);
    if (with_inverse && !plan_inverse(nb, fill, post, pair)) return GPIMHIP_E_BADARG;
    int64_t n = 0;
    auto emit = [&](int launch, const std::vector<TileDesc>& v) {
        for (const TileDesc& t : v) {
            if (out && n < cap) {
                int32_t* r = out + 6 * n;
                r[0] = launch; r[1] = t.ci; r[2] = t.cj; r[3] = t.kb0; r[4] = t.kb1 & 0xffff; r[5] = t.kb1 >> 16;
            }
            ++n;
        }
    };
    for (int j = 0; j < nb; ++j) emit(j, fill[j]);
    for (size_t q = 0; q < post.size(); ++q) emit(nb + (int)q, post[q]);
    *n_out = n;
    return GPIMHIP_OK;
}

int step_plan_ensure(gpimhip_ctx* h, int nb) { return step_plan_build(h, nb, h->splan, false); }
int step_plan_ensure_inv(gpimhip_ctx* h, int nb) { return step_plan_build(h, nb, h->splan_inv, true); }

void step_plan_release(gpimhip_ctx* h) {
    for (StepPlan* P : {&h->splan, &h->splan_inv}) {
        if (P->d_tiles) (void)hipFree(P->d_tiles);
        P->d_tiles = nullptr;
        P->nb = 0;
    }
}

static GemmArgs nt_update(double* A, int64_t ld, const TileDesc* tiles, int n, int64_t rows) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = ld; g.B = A; g.ldb = ld; g.C = A; g.ldc = ld;
    g.alpha = -1.0; g.beta = 1.0; g.tiles = tiles; g.ntiles = n;
    g.sA = g.sB = g.sC = rows * ld;
    return g;
}

// one launch of the step kernel: `potf2` = with the factorisation role for block a.kblk, n hosted tiles in shape q.
// Batches: both roles on all problems of the handle's batch, or (split != nullptr) the factorisation role on problems
// split[0] .. +split[1] and the hosted tiles on problems split[2] .. +split[3], problem index in blockIdx.x.
static int launch_step(gpimhip_ctx* h, StepArgs& a, bool potf2, int n, int q, const int* split = nullptr, bool two_per_cu = false) {
    const int B = h->nbatch;
    a.potf2 = potf2 ? 1 : 0;
    if (!potf2 && n == 0) return GPIMHIP_OK;
    a.pb_off = a.hb_off = 0; a.pb_cnt = a.hb_cnt = B; a.swap = 0;
    dim3 grid(n ? 8 + q * n : 1, B);
    bool alone = potf2;
    // A launch whose UPDATE quadrants need two rounds of the chip at one workgroup per CU and are deep (pair_rule: the
    // second and third panel of a mid-size matrix, most launches around N = 6000) is shorter at two per CU with the
    // factorisation role stretched to ~48 us (N = 4212: 2.72 -> 2.70 ms per iteration); the planner knows (StepPlan::pair).
    if (q == 4 && two_per_cu) alone = false;
    if (split) {
        a.pb_off = split[0]; a.pb_cnt = potf2 ? split[1] : 0; a.hb_off = split[2]; a.hb_cnt = n ? split[3] : 0; a.swap = 1;
        grid = dim3(std::max(a.pb_cnt, a.hb_cnt), n ? 8 + q * n : 1);
        alone = false;          // the hosted half-batch fills the chip: two workgroups per CU
        if (grid.x == 0) return GPIMHIP_OK;
        if (grid.y > 65535) { gpim_set_error("step launch: tile list too long for a split batch"); return GPIMHIP_E_BADARG; }
    }
    if (a.g.rag && n) {           // ragged last block (GemmArgs::rag): the instantiations whose hosted tiles skip its padding
        if (q == 4 && alone) hipLaunchKernelGGL((chol_step_kernel<64, 64, true, true>), grid, dim3(NTH), 0, h->stream, a);
        else if (q == 4) hipLaunchKernelGGL((chol_step_kernel<64, 64, false, true>), grid, dim3(NTH), 0, h->stream, a);
        else if (q == 2) hipLaunchKernelGGL((chol_step_kernel<128, 64, false, true>), grid, dim3(NTH), 0, h->stream, a);
        else hipLaunchKernelGGL((chol_step_kernel<128, 128, false, true>), grid, dim3(NTH), 0, h->stream, a);
    } else if (q == 4 && alone) hipLaunchKernelGGL((chol_step_kernel<64, 64, true>), grid, dim3(NTH), 0, h->stream, a);
    else if (q == 4) hipLaunchKernelGGL((chol_step_kernel<64, 64, false>), grid, dim3(NTH), 0, h->stream, a);
    else if (q == 2) hipLaunchKernelGGL((chol_step_kernel<128, 64, false>), grid, dim3(NTH), 0, h->stream, a);
    else hipLaunchKernelGGL((chol_step_kernel<128, 128, false>), grid, dim3(NTH), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// F_j and D_j for the problems b_off .. b_off + cnt - 1 of the batch
static int launch_solve_diag(gpimhip_ctx* h, double* A, int64_t ld, int j, int nb, int ndiag, int b_off, int cnt,
                             bool fused = false) {
    if (j + 1 >= nb || cnt <= 0) return GPIMHIP_OK;
    // the step launch of column j left a copy of A[j+1, j] in h->pcopy (TILE_COPY).  Not for more than four problems at a time:
    // the workgroups of a diagonal block solve strips redundantly -- free on the chain of one problem, but a lock-step batch
    // of eight is bound by throughput (C3 in four batches of 16: 0.967 s with the 16-row blocks, 0.902 with the 32-row ones, against 0.897; its per-rank share, batches
    // of 4: 0.216 against 0.220).  16-row blocks while the launch fits one round of the chip (137 registers: one 512-thread
    // workgroup per CU), 32-row blocks beyond.
    if (fused && cnt <= 4) {
        const int below = nb - j - 2;
        if ((36 + 4 * below) * cnt <= 256)
            hipLaunchKernelGGL(panel_solve_diag_kernel<16>, dim3(36 + 4 * below, cnt), dim3(512), 0, h->stream, A, ld, j, nb,
                               (const double*)h->dinvB, (const double*)h->pcopy, b_off);
        else
            hipLaunchKernelGGL(panel_solve_diag_kernel<32>, dim3(10 + 2 * below, cnt), dim3(512), 0, h->stream, A, ld, j, nb,
                               (const double*)h->dinvB, (const double*)h->pcopy, b_off);
        HIP_TRY(hipGetLastError());
        return GPIMHIP_OK;
    }
    if (cnt > 4)        // (C3, four concurrent batches of 16: 0.924 -> 0.912 s; same bits)
        hipLaunchKernelGGL(panel_solve_kernel<4>, dim3(4 * (nb - j - 1), cnt), dim3(256), 0, h->stream, A, ld, j, nb,
                           (const double*)h->dinvB, b_off);
    else
        hipLaunchKernelGGL(panel_solve_kernel<2>, dim3(4 * (nb - j - 1), cnt), dim3(256), 0, h->stream, A, ld, j, nb,
                           (const double*)h->dinvB, b_off);
    HIP_TRY(hipGetLastError());
    if (ndiag > 0) {
        hipLaunchKernelGGL(diag_update_kernel, dim3(10 * ndiag, cnt), dim3(256), 0, h->stream, A, ld, j, nb, b_off);
        HIP_TRY(hipGetLastError());
    }
    return GPIMHIP_OK;
}

// Lower Cholesky of the np x np matrix A (np = nb * 128), in place, on h->stream; h->dinvB / h->logdet_part receive the
// inverses of the diagonal blocks (MFMA operand order) and the log-determinant partials.
// Tm == nullptr: A <- L, h->dinv <- the inverses of the diagonal blocks (gpimhip_potrf, the distributed driver).
// Tm != nullptr: A <- L^-1 (every training iteration and prediction inverts the factor right away, gpr.py:192-193,248):
//                the tile operations of the triangular inverse ride in the step launches (plan_inverse), Tm is the
//                np x np temporary of its T phases.
int launch_potrf_steps_f32(gpimhip_ctx* h, double* A, int64_t np, int64_t ld, int32_t* info);   // cholstep32.hip
int launch_potrf_steps(gpimhip_ctx* h, double* A, int64_t np, int64_t ld, int32_t* info, double* Tm, int rag) {
    if (h->fp32) return launch_potrf_steps_f32(h, A, np, ld, info);
    const int nb = (int)(np / NB);
    if (Tm) GP_TRY(step_plan_ensure_inv(h, nb));
    else GP_TRY(step_plan_ensure(h, nb));
    const StepPlan& P = Tm ? h->splan_inv : h->splan;
    const int B = h->nbatch;
    const int host_max_batch = 4;      // (2 / 8 measure the same on C3; 16, i.e. batches of 16 un-split: 0.998 vs 0.907 s)
    StepArgs a;
    a.A = A; a.ld = ld; a.nb = nb; a.Tm = Tm;
    a.dinv_all = h->dinv; a.dinvB_all = h->dinvB; a.logdet = h->logdet_part; a.info = info;
    a.col_off = 0;
    a.Pcopy = h->pcopy;
    // F_j + D_j as one launch where the step launch left the copy it reads (not for the ragged last block row, whose
    // padding rows the hosted tile skips)
    const bool no_fused = getenv("GPIMHIP_NO_FUSED_FD") != nullptr;     // (run-time knob: F_j and D_j as two launches)
    auto fused = [&](int j) { return !no_fused && h->pcopy && P.copy[j] && P.diag[j].n == 1 && !(rag && j + 2 == nb); };
    {
        StageTimer t(h, 0);
        auto hosted_list = [&](int j, int problems, int& q) {
            const int nf = P.fill[j].n;
            a.g = nt_update(A, ld, P.d_tiles + P.fill[j].off, nf, h->np);
            // mixed k-ranges (column updates are up to 2W-1 blocks deep, bulk tiles W or 2W): deal the list to the XCDs in chunks
            a.g.chunk = std::max(1, std::min(64, nf / 512));
            a.g.rag = rag;
            // a large batch is bound by throughput: every hosted k-block counts (P.n_all), not only the trailing update's
            // (batches: 700 / 2800 re-measured on C3 against 300, 2000, 4000 / 1500, 6000 -- the defaults stay)
            q = B > host_max_batch ? host_shape((int64_t)P.n_all[j] * problems, BATCH_QUAD_MAX) : host_shape((int64_t)P.n_update[j] * problems);
            return nf;
        };
        if (B > host_max_batch) {
            // Large batches saturate the chip by themselves, and a launch with the factorisation role alone (one workgroup
            // per problem, 35-50 us) leaves most of it idle.  The two halves of the batch take turns: one launch holds the
            // factorisation role of block column j for one half and the pending tile operations of the OTHER half, whose
            // chain is half a step behind.  Same tile operations in the same order per output element as the hosted form
            // of a stand-alone problem, hence the same bits.
            const int c0 = (B + 1) / 2, c1 = B - c0;
            int q;
            int nf = hosted_list(0, c0, q);
            const int first[4] = {0, 0, 0, c0};
            GP_TRY(launch_step(h, a, false, nf, q, first));                    // (empty today: nothing is pending at step 0)
            for (int j = 0; j < nb; ++j) {
                a.kblk = j;
                nf = hosted_list(j, c1, q);
                const int sa[4] = {0, c0, c0, c1};                             // factor half 0 || pending tiles of half 1
                GP_TRY(launch_step(h, a, true, nf, q, sa));
                GP_TRY(launch_solve_diag(h, A, ld, j, nb, P.diag[j].n, 0, c0, fused(j)));
                nf = j + 1 < nb ? hosted_list(j + 1, c0, q) : 0;
                const int sb[4] = {c0, c1, 0, c0};                             // factor half 1 || pending tiles of half 0, next step
                GP_TRY(launch_step(h, a, true, nf, q, sb));
                GP_TRY(launch_solve_diag(h, A, ld, j, nb, P.diag[j].n, c0, c1, fused(j)));
            }
        } else {
            for (int j = 0; j < nb; ++j) {
                a.kblk = j;
                int q;
                const int nf = hosted_list(j, B, q);
                GP_TRY(launch_step(h, a, true, nf, q, nullptr, P.pair[j] != 0));
                GP_TRY(launch_solve_diag(h, A, ld, j, nb, P.diag[j].n, 0, B, fused(j)));
            }
        }
    }
    if (Tm) {
        StageTimer t(h, 1);
        a.kblk = -1;
        for (const PlanRange& r : P.post) {
            a.g = nt_update(A, ld, P.d_tiles + r.off, r.n, h->np);
            a.g.chunk = std::max(1, std::min(64, r.n / 512));
            a.g.rag = rag;
            // quadrants up to 320 tiles: these launches hold a node's deepest tiles (k-depth up to the node's size), and the
            // launch lasts as long as its longest workgroup (N = 4212: 0.369 -> 0.330 ms against 128; N = 8192 unchanged)
            const int64_t nt = (int64_t)r.n * B;
            // (halves up to 1000 tiles: the two 1024-tile launches of 32-deep tiles at N = 16384 are exactly two rounds of
            // whole tiles -- 1.18 -> 1.09 ms each)
            GP_TRY(launch_step(h, a, false, r.n, nt <= 320 ? 4 : (nt <= 1000 ? 2 : 1)));
        }
    }
    return GPIMHIP_OK;
}

// The chain of one outer panel [p0, p1) whose columns have already received every update from the panels left of
// it (distributed factorisation: right-looking across panels, gpimhip_dist_panel_factor).  Same launches as above
// with the left-looking window restricted to the panel: colfill[j - p0] = tiles (i, j), i > j, k-blocks [p0, j).
// A is addressed with GLOBAL block indices (the caller shifts the base of a local column panel accordingly).
int launch_panel_chain(gpimhip_ctx* h, double* A, int64_t ld, int p0, int p1, int nb, const TileDesc* tiles,
                       const PlanRange* colfill, int32_t* info) {
    const int B = h->nbatch;
    StepArgs a;
    a.A = A; a.ld = ld; a.nb = nb; a.Tm = nullptr;
    a.dinv_all = h->dinv; a.dinvB_all = h->dinvB; a.logdet = h->logdet_part; a.info = info;
    a.col_off = 0;
    a.Pcopy = nullptr;
    for (int j = p0; j < p1; ++j) {
        a.kblk = j;
        const PlanRange f = colfill[j - p0];
        a.g = nt_update(A, ld, tiles + f.off, f.n, h->np);
        a.g.chunk = 1;
        GP_TRY(launch_step(h, a, true, f.n, f.n <= 128 ? 4 : 2));
        GP_TRY(launch_solve_diag(h, A, ld, j, nb, std::max(0, p1 - j - 1), 0, B));
    }
    return GPIMHIP_OK;
}
