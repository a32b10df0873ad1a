// gemm_body.hpp -- the fp64 MFMA tile engine of libgpimhip as a device function (see gemm.hip for the launch side).
#pragma once
#include "common.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define LD_MK (GEMM_BK + 2)
// doubles between two 8-row groups of a directly staged k-contiguous tile.  128 = the groups back to back: rows m and
// m + 8 of a fragment read then share their banks (2-way conflict on every read); 130 shifts the second group of a
// 16-row fragment by 16 bytes, onto the banks the first one leaves free (within a group the XOR swizzle uses the
// chunks of one parity of k >> 1 per row parity) -- conflict-free, and 16 x 130 <= 18 x 128 still fits a stage.
#ifndef MK_GROUP
#define MK_GROUP 130
#endif

// TS = tile side handled by one workgroup (128 or 64); NT = threads.  One staged operand tile is
// TS x 16 doubles = 8*TS 16-byte chunks whatever its orientation.
template <bool KM, int NT, int TS>
__device__ __forceinline__ void stage_load(d2 (&r)[8 * TS / NT], const double* __restrict__ base, int64_t ld,
                                           int64_t mrow0, int64_t kcol0, int tid) {
    // MK: tile element (m,k) lives at base[(mrow0+m)*ld + kcol0 + k]
    // KM: tile element (k,m) lives at base[(kcol0+k)*ld + mrow0 + m]
#pragma unroll
    for (int i = 0; i < 8 * TS / NT; ++i) {
        const int c = tid + NT * i;
        if (!KM) {
            const int row = c >> 3, c16 = c & 7;
            r[i] = *reinterpret_cast<const d2*>(base + (mrow0 + row) * ld + kcol0 + c16 * 2);
        } else {
            const int krow = c / (TS / 2), c16 = c % (TS / 2);
            r[i] = *reinterpret_cast<const d2*>(base + (kcol0 + krow) * ld + mrow0 + c16 * 2);
        }
    }
}

template <bool KM, int NT, int TS>
__device__ __forceinline__ void stage_store(const d2 (&r)[8 * TS / NT], double* lds, int tid) {
#pragma unroll
    for (int i = 0; i < 8 * TS / NT; ++i) {
        const int c = tid + NT * i;
        if (!KM) {
            const int row = c >> 3, c16 = c & 7;
            *reinterpret_cast<d2*>(lds + row * LD_MK + c16 * 2) = r[i];
        } else {
            const int krow = c / (TS / 2), c16 = c % (TS / 2);
            *reinterpret_cast<d2*>(lds + krow * (TS + 16) + c16 * 2) = r[i];
        }
    }
}

// Direct global -> LDS staging (global_load_lds_dwordx4) of an m-contiguous ("KM") operand tile with
// TS = 128: one k-row is 128 doubles = 1 KB = one wave-wide 16-byte load, written by the hardware to
// lds[krow][lane*2 .. lane*2+1] without passing through VGPRs -- no ds_write, no register staging.
// Tracked by vmcnt; the caller waits for vmcnt(0) before the barrier that publishes the stage.
// (tools/gemm_ablate.hip: 66.1 -> 68.5 TFLOP/s for the loop with both operands staged this way.)
template <int NW>
__device__ __forceinline__ void stage_direct_km(const double* __restrict__ base, int64_t ld, int64_t mrow0,
                                                int64_t kcol0, double* lds, int wave, int lane) {
    constexpr int ROWS = GEMM_BK / NW;      // k-rows per wave
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int krow = wave * ROWS + i;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(base + (kcol0 + krow) * ld + mrow0 + lane * 2),
            (__attribute__((address_space(3))) void*)(lds + krow * (128 + 16)), 16, 0, 0);
    }
}

// The same for a k-contiguous ("MK") operand tile with TS = 128 rows of 16 doubles (128 B): one wave-wide
// load fetches 8 rows x 8 sixteen-byte chunks (fully coalesced: whole 128-byte lines) and lands as one
// contiguous 1 KB group in LDS.  LDS position p = lane of the group holds row p>>3, chunk (p&7) ^ (p>>3):
// the XOR swizzle spreads the fragment reads over the banks (element (m,k) sits at
// (m>>3)*128 + ((m&7)*8 + ((k>>1) ^ (m&7)))*2 + (k&1); two rows 8 apart share a bank, nothing worse --
// 0.4 TFLOP/s in tools/gemm_ablate.hip against the padded [128][18] layout it replaces).
template <int NW, int TS>
__device__ __forceinline__ void stage_direct_mk(const double* __restrict__ base, int64_t ld, int64_t mrow0,
                                                int64_t kcol0, double* lds, int wave, int lane) {
    constexpr int GROUPS = (TS / 8) / NW;   // 8-row groups per wave
    const int r8 = lane >> 3, c8 = (lane & 7) ^ r8;
#pragma unroll
    for (int i = 0; i < GROUPS; ++i) {
        const int q = wave * GROUPS + i;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(base + (mrow0 + q * 8 + r8) * ld + kcol0 + c8 * 2),
            (__attribute__((address_space(3))) void*)(lds + q * MK_GROUP), 16, 0, 0);
    }
}
__device__ __forceinline__ double frag_mk_swz(const double* lds, int m0, int kk, int lane) {
    const int m = m0 + (lane & 15), k = kk * 4 + (lane >> 4);
    return lds[(m >> 3) * MK_GROUP + ((m & 7) * 8 + ((k >> 1) ^ (m & 7))) * 2 + (k & 1)];
}

// Fragments of TWO k-steps per LDS instruction.  The four k-steps of a 16-deep stage may visit its 16 k-values in any order
// as long as both operands use the same one; with  kappa(s, kq) = 8 (s >> 1) + 2 kq + (s & 1)  (k-step s, kq = lane >> 4)
// a lane's values for the steps 2h and 2h + 1 are the aligned pair k = 8 h + 2 kq, + 1 -- one 16-byte chunk of a
// k-contiguous ("MK") operand row, i.e. one ds_read_b128 instead of two ds_read_b64 (the swizzle keeps a chunk whole).
// An LDS instruction costs its SIMD 10-25 cycles of MFMA issue whichever wave it comes from (tools/r5_trail_probe.hip);
// the 8-wave shapes hosted by the Cholesky step kernel issue 0.75-1.5 fragment reads per MFMA.  m-contiguous ("KM")
// operands follow the same k order with single reads.
__device__ __forceinline__ d2 frag2_mk_swz(const double* lds, int m0, int h, int lane) {
    const int m = m0 + (lane & 15), kq = lane >> 4;
    return *reinterpret_cast<const d2*>(lds + (m >> 3) * MK_GROUP + ((m & 7) * 8 + ((4 * h + kq) ^ (m & 7))) * 2);
}
template <int TS>
__device__ __forceinline__ d2 frag2_km(const double* lds, int m0, int h, int lane) {
    const int k = 8 * h + 2 * (lane >> 4), m = m0 + (lane & 15);
    return (d2){lds[k * (TS + 16) + m], lds[(k + 1) * (TS + 16) + m]};
}
template <bool KM, int TS>
__device__ __forceinline__ d2 frag2(const double* lds, int m0, int h, int lane) {
    if (KM) return frag2_km<TS>(lds, m0, h, lane);
    return frag2_mk_swz(lds, m0, h, lane);
}

template <bool KM, int NW, int TS>
__device__ __forceinline__ void stage_direct(const double* __restrict__ base, int64_t ld, int64_t mrow0, int64_t kcol0,
                                             double* lds, int wave, int lane) {
    if (KM) stage_direct_km<NW>(base, ld, mrow0, kcol0, lds, wave, lane);
    else stage_direct_mk<NW, TS>(base, ld, mrow0, kcol0, lds, wave, lane);
}

template <bool KM, int TS>
__device__ __forceinline__ double frag(const double* lds, int m0, int kk, int lane) {
    // element (m = m0 + (lane&15), k = kk*4 + (lane>>4)) of the staged tile; both layouts are
    // bank-conflict free for ds_read_b64: [TS][18] (18 % 32 == 18 -> rows spread), [16][TS+16]
    if (!KM) return lds[(m0 + (lane & 15)) * LD_MK + kk * 4 + (lane >> 4)];
    return lds[(kk * 4 + (lane >> 4)) * (TS + 16) + m0 + (lane & 15)];
}

// NW = waves per workgroup, TSM x TSN = the part of a 128x128 tile one workgroup computes:
//   (4, 128, 128)  2x2 waves of 64x64: bulk launches, two workgroups share a CU
//   (8, 128, 128)  4x2 waves of 32x64: at most one tile per CU -> still two MFMA waves per SIMD (a
//                  lone fp64-MFMA wave issues only every ~140 cycles)
//   (4,  64,  64)  2x2 waves of 32x32 on a quadrant: few tiles, each spread over four CUs
//   (8,  64, 128)  4x2 waves of 16x64 on a row half: in-place panel solves (a workgroup must own
//                  whole rows of the tile it overwrites), each tile spread over two CUs
// bx / by: the workgroup's position in a plain tile launch (block index x / y); smem: gemm_smem_doubles<TSM, TSN>()
// doubles of LDS.  A device function so that other kernels can host tile work next to their own (cholstep.hip:
// the diagonal-block factorisation shares its launch with trailing-update tiles).
// NSTG: LDS stages.  2 = double buffering (prefetch distance 1; with two or more workgroups per CU the other
// workgroup's MFMAs cover a stage's load latency).  > 2 = a ring with prefetch distance NSTG - 1 for launches that
// can only have ONE workgroup per CU (tiles hosted by the Cholesky step kernel, whose factorisation role sizes the
// launch's LDS): the loads of NSTG - 1 k-steps are in flight while one is consumed.
template <int TSM, int TSN, int NSTG = 2>
__host__ __device__ constexpr int gemm_smem_doubles() {
    constexpr int TSX = TSM > TSN ? TSM : TSN;
    return 2 * NSTG * ((TSX * LD_MK > GEMM_BK * (TSX + 16)) ? TSX * LD_MK : GEMM_BK * (TSX + 16));
}
// XCD-aware bijective remap of a launch's workgroup index to a position of its tile list (block b runs on XCD
// b % 8, in order b / 8 on that XCD).
//  chunk == 0: every XCD gets one contiguous slice of the tile list -- best L2 reuse when all
//              tiles cost the same (SYRK-shaped trailing updates).
//  chunk  > 0: the list is dealt to the XCDs in chunks of that many tiles (one 8x8 patch), back
//              and forth, so lists sorted by decreasing k-range stay balanced across XCDs.
template <int TSM, int TSN>
__device__ __forceinline__ int gemm_tile_pos(int n, int chunk, int bx, int& quad) {
    constexpr int QUADS = (128 / TSM) * (128 / TSN);    // workgroups per 128x128 tile
    const int b = bx / QUADS;
    quad = bx % QUADS;
    if (QUADS > 1) return b;
    if (chunk == 0) {
        const int q = n >> 3, r = n & 7, x = b & 7, yy = b >> 3;
        return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + yy;
    }
    const int C = chunk, full = (n / (8 * C)) * (8 * C);
    if (b < full) {
        // serpentine: odd rounds deal in reverse, so that on a list sorted by cost no XCD always
        // gets the most expensive chunk of the round (16 % spread between XCD 0 and 7 otherwise)
        const int x = b & 7, y = b >> 3, round = y / C;
        return (round * 8 + ((round & 1) ? 7 - x : x)) * C + (y % C);
    }
    return b;
}

// RAG: the launch has a ragged last block (GemmArgs::rag != 0).  A template parameter so that the k-loop of every other
// launch keeps its schedule (as a run-time test inside the loop it cost the K^-1 product at N = 16384 0.4 %).
template <bool A_KM, bool B_KM, int EPI, int NW, int TSM, int TSN, int NSTG = 2, bool RAG = false>
__device__ __forceinline__ void gemm_tile_core(GemmArgs g, TileDesc t, const int quad, const int by, double* __restrict__ smem);

template <bool A_KM, bool B_KM, int EPI, int NW, int TSM, int TSN, int NSTG = 2>
__device__ __forceinline__ void gemm_tile_body(GemmArgs g, const int bx, const int by, double* __restrict__ smem) {
    int quad;
    const int p = gemm_tile_pos<TSM, TSN>(g.ntiles, g.chunk, bx, quad);
    TileDesc t;
    if (g.rect_cols > 0) {
        const int per = 8 * g.rect_cols, s = p / per, q = p - s * per, rows_here = min(8, g.rect_rows - 8 * s);
        t.ci = 8 * s + q % rows_here;
        t.cj = q / rows_here;
        t.kb0 = g.kfix0;
        t.kb1 = g.kfix1;
    } else {
        t = g.tiles[p];
    }
    if (g.rag) gemm_tile_core<A_KM, B_KM, EPI, NW, TSM, TSN, NSTG, true>(g, t, quad, by, smem);
    else gemm_tile_core<A_KM, B_KM, EPI, NW, TSM, TSN, NSTG, false>(g, t, quad, by, smem);
}

// one tile (or its quadrant / half `quad`) of a tile-engine launch
template <bool A_KM, bool B_KM, int EPI, int NW, int TSM, int TSN, int NSTG, bool RAG>
__device__ __forceinline__ void gemm_tile_core(GemmArgs g, TileDesc t, const int quad, const int by, double* __restrict__ smem) {
    constexpr int NT = NW * 64;             // threads
    constexpr int WGM = NW / 2;             // waves along m (2 along n)
    constexpr int WROWS = TSM / WGM;        // rows per wave
    constexpr int WCOLS = TSN / 2;          // cols per wave
    constexpr int MT = WROWS / 16;          // 16x16 accumulators per wave: MT x NTL
    constexpr int NTL = WCOLS / 16;
    constexpr int TSX = TSM > TSN ? TSM : TSN;
    constexpr int STAGE = (TSX * LD_MK > GEMM_BK * (TSX + 16)) ? TSX * LD_MK : GEMM_BK * (TSX + 16);
    constexpr int NCHA = 8 * TSM / NT, NCHB = 8 * TSN / NT;   // 16-byte chunks per thread per stage
    // staged straight into LDS: every 128-wide operand tile, and 64-wide k-contiguous ones (a 64-wide
    // m-contiguous k-row is only half a wave-wide load)
    constexpr bool ADIR = TSM == 128 || !A_KM, BDIR = TSN == 128 || !B_KM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    constexpr int QN = 128 / TSN;                       // workgroups per tile along n
    if (g.cj_max > 0 && t.cj >= g.cj_max) return;   // (the whole workgroup: t is uniform)
    const int ccb = g.cmap ? t.kb0 : t.cj;          // block column of the output (see GemmArgs::cmap)
    if (g.kfix1 > g.kfix0) { t.kb0 = g.kfix0; t.kb1 = g.kfix1; }
    int nsteps = (t.kb1 - t.kb0) * (NB / GEMM_BK);
    const int qi = (quad / QN) * TSM, qj = (quad % QN) * TSN;
    // ragged last block (GemmArgs::rag): its rows / columns >= 64 are identity padding
    if (RAG && t.kb1 == g.rag) nsteps -= 64 / GEMM_BK;
    const bool rag_row = RAG && t.ci == g.rag - 1;
    if (RAG && TSM <= 64 && rag_row && qi >= 64) return;            // (the whole workgroup)
    const bool dead = RAG && rag_row && qi + wm * WROWS >= 64;      // this wave's rows: no MFMAs, nothing stored
    // batch: by selects the problem; operands advance by their per-problem strides
    g.A += by * g.sA;
    g.B += by * g.sB;
    if (g.C) g.C += by * g.sC;
    if (g.colpart) g.colpart += by * g.sColpart;

    // operand origins (element units)
    const int64_t a_m0 = (int64_t)(t.ci + (A_KM ? g.a_coff : g.a_roff)) * NB + qi;
    const int64_t a_k0 = (int64_t)(t.kb0 + (A_KM ? g.a_roff : g.a_coff)) * NB;
    const int64_t b_n0 = (int64_t)(t.cj + (B_KM ? g.b_coff : g.b_roff)) * NB + qj;
    const int64_t b_k0 = (int64_t)(t.kb0 + (B_KM ? g.b_roff : g.b_coff)) * NB;

    d4 acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

    // krev: walk the k-range from its END.  Tiles of one patch whose ranges share their upper end
    // (K^-1 = L^-T L^-1: [ci, nb)) then sweep the same operand rows at the same time, so the panels
    // they share are still in the XCD's L2 when the next tile asks for them.
    const int64_t kfirst = g.krev ? (int64_t)(nsteps - 1) * GEMM_BK : 0;
    const int64_t kstride = g.krev ? -GEMM_BK : GEMM_BK;
    if constexpr (NSTG > 2) {
        static_assert(ADIR && BDIR, "the stage ring needs both operands staged global -> LDS directly");
        constexpr int DIST = NSTG - 1;                                  // prefetch distance in k-steps
        constexpr int LPS = (A_KM ? GEMM_BK / NW : (TSM / 8) / NW) + (B_KM ? GEMM_BK / NW : (TSN / 8) / NW);   // loads per wave per stage
        static_assert(LPS * (DIST - 1) < 64, "vmcnt is a 6-bit counter");
        auto issue = [&](int st) {
            double* An = smem + (st % NSTG) * 2 * STAGE;
            const int64_t koff = kfirst + (int64_t)st * kstride;
            stage_direct<A_KM, NW, TSM>(g.A, g.lda, a_m0, a_k0 + koff, An, wave, lane);
            stage_direct<B_KM, NW, TSN>(g.B, g.ldb, b_n0, b_k0 + koff, An + STAGE, wave, lane);
        };
        for (int st = 0; st < DIST && st < nsteps; ++st) issue(st);
        for (int s = 0; s < nsteps; ++s) {
            // stage s has landed once at most the DIST - 1 younger stages are still in flight (loads retire in order)
            if (s + DIST - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS * (DIST - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();        // ... for every wave; and every wave is done with stage s - 1, whose slot is refilled now
            if (s + DIST < nsteps) issue(s + DIST);
            const double* As = smem + (s % NSTG) * 2 * STAGE;
            const double* Bs = As + STAGE;
            __builtin_amdgcn_s_setprio(3);
            if (!RAG || !dead) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                d2 a[MT], bb[NTL];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = frag2<A_KM, TSM>(As, wm * WROWS + i * 16, h, lane);
#pragma unroll
                for (int j = 0; j < NTL; ++j) bb[j] = frag2<B_KM, TSN>(Bs, wn * WCOLS + j * 16, h, lane);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NTL; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][e], bb[j][e], acc[i][j], 0, 0, 0);
            }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
    } else {
        d2 ra[NCHA], rb[NCHB];
        if (nsteps > 0) {
            if (ADIR) stage_direct<A_KM, NW, TSM>(g.A, g.lda, a_m0, a_k0 + kfirst, smem, wave, lane);
            else stage_load<A_KM, NT, TSM>(ra, g.A, g.lda, a_m0, a_k0 + kfirst, tid);
            if (BDIR) stage_direct<B_KM, NW, TSN>(g.B, g.ldb, b_n0, b_k0 + kfirst, smem + STAGE, wave, lane);
            else stage_load<B_KM, NT, TSN>(rb, g.B, g.ldb, b_n0, b_k0 + kfirst, tid);
            if (!ADIR) stage_store<A_KM, NT, TSM>(ra, smem, tid);
            if (!BDIR) stage_store<B_KM, NT, TSN>(rb, smem + STAGE, tid);
            if (ADIR || BDIR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();

        for (int s = 0; s < nsteps; ++s) {
            const double* As = smem + (s & 1) * 2 * STAGE;
            const double* Bs = As + STAGE;
            const bool more = (s + 1 < nsteps);
            // 4-wave shapes: the fragments of the first MFMA group are requested BEFORE the next stage's global loads are
            // issued, so their LDS latency passes behind those ~80 cycles of address arithmetic and load issue (K^-1
            // product at N = 16384: 21.68 -> 21.51 ms; the 8-wave workgroups hosted by the Cholesky step kernel lose 2 %
            // with it -- they run at the 128-register limit)
            constexpr bool HOIST = NW == 4;
            double a0[MT], b0[NTL];
            if constexpr (HOIST) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    a0[i] = (ADIR && !A_KM) ? frag_mk_swz(As, wm * WROWS + i * 16, 0, lane)
                                            : frag<A_KM, TSM>(As, wm * WROWS + i * 16, 0, lane);
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    b0[j] = (BDIR && !B_KM) ? frag_mk_swz(Bs, wn * WCOLS + j * 16, 0, lane)
                                            : frag<B_KM, TSN>(Bs, wn * WCOLS + j * 16, 0, lane);
            }
            if (more) {
                // the other stage buffer was last read in step s-1: every wave is past that barrier
                double* An = smem + ((s + 1) & 1) * 2 * STAGE;
                const int64_t koff = kfirst + (int64_t)(s + 1) * kstride;
                if (ADIR) stage_direct<A_KM, NW, TSM>(g.A, g.lda, a_m0, a_k0 + koff, An, wave, lane);
                else stage_load<A_KM, NT, TSM>(ra, g.A, g.lda, a_m0, a_k0 + koff, tid);
                if (BDIR) stage_direct<B_KM, NW, TSN>(g.B, g.ldb, b_n0, b_k0 + koff, An + STAGE, wave, lane);
                else stage_load<B_KM, NT, TSN>(rb, g.B, g.ldb, b_n0, b_k0 + koff, tid);
            }
            // The MFMA block runs at raised wave priority: the arbiter then prefers this wave's MFMAs and
            // fragment reads over the other resident wave's staging instructions, which otherwise steal
            // issue slots from the matrix pipe (tools/gemm_ablate.hip: 66.1 -> 71.2 TFLOP/s for this loop).
            __builtin_amdgcn_s_setprio(3);
            if (!RAG || !dead) {
            // (pairs for the 8-wave shapes only: the 4-wave shapes issue 0.5 fragment reads per MFMA and measured 0.2 % slower
            // with them at N = 16384 -- K^-1 product, variance product, the launches after the last step)
            if constexpr (NW != 8) {
    #pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double a[MT], bb[NTL];
    #pragma unroll
                for (int i = 0; i < MT; ++i)
                    a[i] = (ADIR && !A_KM) ? frag_mk_swz(As, wm * WROWS + i * 16, kk, lane)
                                           : frag<A_KM, TSM>(As, wm * WROWS + i * 16, kk, lane);
    #pragma unroll
                for (int j = 0; j < NTL; ++j)
                    bb[j] = (BDIR && !B_KM) ? frag_mk_swz(Bs, wn * WCOLS + j * 16, kk, lane)
                                            : frag<B_KM, TSN>(Bs, wn * WCOLS + j * 16, kk, lane);
                if (HOIST && kk == 0) {
    #pragma unroll
                    for (int i = 0; i < MT; ++i) a[i] = a0[i];
    #pragma unroll
                    for (int j = 0; j < NTL; ++j) bb[j] = b0[j];
                }
    #pragma unroll
                for (int i = 0; i < MT; ++i)
    #pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bb[j], acc[i][j], 0, 0, 0);
            }
            } else {
    #pragma unroll
            for (int h = 0; h < 2; ++h) {
                d2 a[MT], bb[NTL];
                if (false) {
    #pragma unroll
                    for (int i = 0; i < MT; ++i) a[i] = a0[i];
    #pragma unroll
                    for (int j = 0; j < NTL; ++j) bb[j] = b0[j];
                } else {
    #pragma unroll
                    for (int i = 0; i < MT; ++i) a[i] = frag2<A_KM, TSM>(As, wm * WROWS + i * 16, h, lane);
    #pragma unroll
                    for (int j = 0; j < NTL; ++j) bb[j] = frag2<B_KM, TSN>(Bs, wn * WCOLS + j * 16, h, lane);
                }
    #pragma unroll
                for (int e = 0; e < 2; ++e)
    #pragma unroll
                    for (int i = 0; i < MT; ++i)
    #pragma unroll
                        for (int j = 0; j < NTL; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][e], bb[j][e], acc[i][j], 0, 0, 0);
            }
            }
            }
            __builtin_amdgcn_s_setprio(0);
            if (more) {
                double* An = smem + ((s + 1) & 1) * 2 * STAGE;
                if (!ADIR) stage_store<A_KM, NT, TSM>(ra, An, tid);
                if (!BDIR) stage_store<B_KM, NT, TSN>(rb, An + STAGE, tid);
                if (ADIR || BDIR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }

    }

    if (EPI == EPI_STORE) {
        if (RAG && dead) return;                // (no barrier after this point)
        const int64_t crow0 = (int64_t)(t.ci + g.c_roff) * NB + qi + wm * WROWS + (lane >> 4);
        const int64_t ccol0 = (int64_t)(ccb + g.c_coff) * NB + qj + wn * WCOLS + (lane & 15);
        const double alpha = g.alpha, beta = g.beta;
        if (beta != 0.0) {
            // accumulate into C: fetch one 16-row band of the wave's sub-tile (NTL x 4 values per lane)
            // with all loads in flight, then combine and store -- element by element the loads and
            // stores serialise into 2 x 64 memory round trips per tile
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                double cv[NTL][4];
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        cv[j][rg] = g.C[(crow0 + i * 16 + 4 * rg) * g.ldc + ccol0 + j * 16];
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const double v = alpha * acc[i][j][rg] + beta * cv[j][rg];
                        g.C[(crow0 + i * 16 + 4 * rg) * g.ldc + ccol0 + j * 16] = v;
                        if (g.C2)       // (uniform: one tile of a launch at most)
                            g.C2[(qi + wm * WROWS + (lane >> 4) + i * 16 + 4 * rg) * NB + qj + wn * WCOLS + (lane & 15) + j * 16] = v;
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        g.C[(crow0 + i * 16 + 4 * rg) * g.ldc + ccol0 + j * 16] = alpha * acc[i][j][rg];
        }
    } else {
        // column sums of squares of the 128x128 product tile -> colpart[ci][cj*128 + col]  (full tiles only)
        double* red = smem;      // [WGM][128]; all waves are past the last barrier of the k-loop
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) s += acc[i][j][rg] * acc[i][j][rg];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) red[wm * 128 + wn * WCOLS + j * 16 + lane] = s;
        }
        __syncthreads();
        if (tid < 128) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < WGM; ++w) tot += red[w * 128 + tid];
            g.colpart[(int64_t)t.ci * g.ld_colpart + (int64_t)(t.cj + g.c_coff) * NB + tid] = tot;
        }
    }
}

