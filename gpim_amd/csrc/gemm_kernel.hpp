// gemm_kernel.hpp -- the tile engine of gemm.hip written once for both element types (kernel template + launch
// dispatch by shape); instantiated for float in gemm32.hip (precision = 'single').  The double kernels of
// gemm.hip are the same algorithm in its original, non-generic form (see the note there).  Design notes: gemm.hip.
#pragma once
#include <stdlib.h>
#include "common.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// The engine is written once for both element types (precision = 'double' / 'single' of the reconstructor,
// gpim/gpreg/gpr.py:104-113).  What depends on the type:
//   CH    16-byte chunk (2 doubles / 4 floats), EPC elements per chunk
//   BK    k-depth of one LDS stage = 8 chunks = 128 bytes of a k-contiguous row (16 doubles / 32 floats), so the
//         byte geometry of every staging scheme is the same for both types
//   MFMA  v_mfma_f64_16x16x4_f64 (64 cycles) / v_mfma_f32_16x16x4_f32 (32 cycles): same A/B lane maps, different
//         C/D row map -- f64: row = (lane>>4) + 4*reg, f32: row = 4*(lane>>4) + reg (cdna_hip_programming.md 3)
template <typename R> struct RT;
template <> struct RT<double> {
    typedef d2 CH; typedef d4 ACC;
    static constexpr int EPC = 2, BK = 16;
    static __device__ __forceinline__ ACC mfma(double a, double b, ACC c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int drow(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct RT<float> {
    typedef f4 CH; typedef f4 ACC;
    static constexpr int EPC = 4, BK = 32;
    static __device__ __forceinline__ ACC mfma(float a, float b, ACC c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int drow(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

// padded row strides of the register-staged LDS layouts (elements): k-contiguous rows BK + one chunk, m-contiguous
// rows TS + 16 -- both spread the four k-rows of a fragment read over distinct banks for either type
#define LD_MK_OF(R) (RT<R>::BK + RT<R>::EPC)

// TS = tile side handled by one workgroup (128 or 64); NT = threads.  One staged operand tile is
// TS x BK elements = 8*TS 16-byte chunks whatever its orientation and type.
template <typename R, bool KM, int NT, int TS>
__device__ __forceinline__ void stage_load(typename RT<R>::CH (&r)[8 * TS / NT], const R* __restrict__ base, int64_t ld,
                                           int64_t mrow0, int64_t kcol0, int tid) {
    // MK: tile element (m,k) lives at base[(mrow0+m)*ld + kcol0 + k]
    // KM: tile element (k,m) lives at base[(kcol0+k)*ld + mrow0 + m]
    typedef typename RT<R>::CH CH;
    constexpr int EPC = RT<R>::EPC;
#pragma unroll
    for (int i = 0; i < 8 * TS / NT; ++i) {
        const int c = tid + NT * i;
        if (!KM) {
            const int row = c >> 3, c16 = c & 7;
            r[i] = *reinterpret_cast<const CH*>(base + (mrow0 + row) * ld + kcol0 + c16 * EPC);
        } else {
            const int krow = c / (TS / EPC), c16 = c % (TS / EPC);
            r[i] = *reinterpret_cast<const CH*>(base + (kcol0 + krow) * ld + mrow0 + c16 * EPC);
        }
    }
}

template <typename R, bool KM, int NT, int TS>
__device__ __forceinline__ void stage_store(const typename RT<R>::CH (&r)[8 * TS / NT], R* lds, int tid) {
    typedef typename RT<R>::CH CH;
    constexpr int EPC = RT<R>::EPC;
#pragma unroll
    for (int i = 0; i < 8 * TS / NT; ++i) {
        const int c = tid + NT * i;
        if (!KM) {
            const int row = c >> 3, c16 = c & 7;
            *reinterpret_cast<CH*>(lds + row * LD_MK_OF(R) + c16 * EPC) = r[i];
        } else {
            const int krow = c / (TS / EPC), c16 = c % (TS / EPC);
            *reinterpret_cast<CH*>(lds + krow * (TS + 16) + c16 * EPC) = r[i];
        }
    }
}

// Direct global -> LDS staging (global_load_lds_dwordx4) of an m-contiguous ("KM") fp64 operand tile with
// TS = 128: one k-row is 128 doubles = 1 KB = one wave-wide 16-byte load, written by the hardware to
// lds[krow][lane*2 .. lane*2+1] without passing through VGPRs -- no ds_write, no register staging.
// Tracked by vmcnt; the caller waits for vmcnt(0) before the barrier that publishes the stage.
// (tools/gemm_ablate.hip: 66.1 -> 68.5 TFLOP/s for the loop with both operands staged this way.)
// (fp32: a k-row is only half a wave-wide load and two rows would land back to back, without the padding that
// keeps fragment reads conflict free -- m-contiguous float tiles go through registers.)
template <int NW>
__device__ __forceinline__ void stage_direct_km(const double* __restrict__ base, int64_t ld, int64_t mrow0,
                                                int64_t kcol0, double* lds, int wave, int lane) {
    constexpr int ROWS = 16 / NW;           // k-rows per wave
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int krow = wave * ROWS + i;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(base + (kcol0 + krow) * ld + mrow0 + lane * 2),
            (__attribute__((address_space(3))) void*)(lds + krow * (128 + 16)), 16, 0, 0);
    }
}

// The same for a k-contiguous ("MK") operand tile of TS rows of 128 bytes (16 doubles / 32 floats): one wave-wide
// load fetches 8 rows x 8 sixteen-byte chunks (fully coalesced: whole 128-byte lines) and lands as one
// contiguous 1 KB group in LDS.  LDS position p = lane of the group holds row p>>3, chunk (p&7) ^ (p>>3):
// the XOR swizzle spreads the fragment reads over the banks (element (m,k) sits in chunk (k / EPC) ^ (m&7) of
// row m&7 of group m>>3; two rows 8 apart share a bank, nothing worse --
// 0.4 TFLOP/s in tools/gemm_ablate.hip against the padded [128][18] layout it replaces).
template <typename R, int NW, int TS>
__device__ __forceinline__ void stage_direct_mk(const R* __restrict__ base, int64_t ld, int64_t mrow0,
                                                int64_t kcol0, R* lds, int wave, int lane) {
    constexpr int GROUPS = (TS / 8) / NW;   // 8-row groups per wave
    constexpr int EPC = RT<R>::EPC;
    const int r8 = lane >> 3, c8 = (lane & 7) ^ r8;
#pragma unroll
    for (int i = 0; i < GROUPS; ++i) {
        const int q = wave * GROUPS + i;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(base + (mrow0 + q * 8 + r8) * ld + kcol0 + c8 * EPC),
            (__attribute__((address_space(3))) void*)(lds + q * (64 * EPC)), 16, 0, 0);
    }
}
template <typename R>
__device__ __forceinline__ R frag_mk_swz(const R* lds, int m0, int kk, int lane) {
    constexpr int EPC = RT<R>::EPC;
    const int m = m0 + (lane & 15), k = kk * 4 + (lane >> 4);
    return lds[(m >> 3) * (64 * EPC) + ((m & 7) * 8 + ((k / EPC) ^ (m & 7))) * EPC + (k % EPC)];
}

template <typename R, bool KM, int NW, int TS>
__device__ __forceinline__ void stage_direct(const R* __restrict__ base, int64_t ld, int64_t mrow0, int64_t kcol0,
                                             R* lds, int wave, int lane) {
    if constexpr (KM) stage_direct_km<NW>(base, ld, mrow0, kcol0, lds, wave, lane);   // fp64 only (see ADIR / BDIR)
    else stage_direct_mk<R, NW, TS>(base, ld, mrow0, kcol0, lds, wave, lane);
}

template <typename R, bool KM, int TS>
__device__ __forceinline__ R frag(const R* lds, int m0, int kk, int lane) {
    // element (m = m0 + (lane&15), k = kk*4 + (lane>>4)) of the staged tile; both layouts are
    // bank-conflict free: [TS][BK + chunk] (rows spread), [BK][TS+16]
    if (!KM) return lds[(m0 + (lane & 15)) * LD_MK_OF(R) + kk * 4 + (lane >> 4)];
    return lds[(kk * 4 + (lane >> 4)) * (TS + 16) + m0 + (lane & 15)];
}

// NW = waves per workgroup, TSM x TSN = the part of a 128x128 tile one workgroup computes:
//   (4, 128, 128)  2x2 waves of 64x64: bulk launches, two workgroups share a CU
//   (8, 128, 128)  4x2 waves of 32x64: at most one tile per CU -> still two MFMA waves per SIMD (a
//                  lone fp64-MFMA wave issues only every ~140 cycles)
//   (4,  64,  64)  2x2 waves of 32x32 on a quadrant: few tiles, each spread over four CUs
//   (8,  64, 128)  4x2 waves of 16x64 on a row half: in-place panel solves (a workgroup must own
//                  whole rows of the tile it overwrites), each tile spread over two CUs
// LDS elements of one workgroup: two stages of an A and a B operand tile each
template <typename R, int TSM, int TSN>
constexpr int gemm_smem_elems_t() {
    constexpr int TSX = TSM > TSN ? TSM : TSN;
    constexpr int BK = RT<R>::BK;
    return 4 * ((TSX * LD_MK_OF(R) > BK * (TSX + 16)) ? TSX * LD_MK_OF(R) : BK * (TSX + 16));
}

// One workgroup's share of a tile launch as a device function: bx / by = the workgroup's position in the launch
// (the block index of gemm_tiles_kernel_t; the step kernel of cholstep32.hip hosts tiles next to a factorisation
// role), smem = gemm_smem_elems_t<R, TSM, TSN>() elements of LDS.
template <typename R, bool A_KM, bool B_KM, int EPI, int NW, int TSM, int TSN>
__device__ __forceinline__ void gemm_tile_body_t(GemmArgs g, const int bx, const int by, R* __restrict__ smem) {
    typedef typename RT<R>::CH CH;
    typedef typename RT<R>::ACC ACC;
    constexpr int BK = RT<R>::BK;           // k-depth of one stage
    constexpr int KSTEPS = BK / 4;          // MFMA k-steps per stage
    constexpr int NT = NW * 64;             // threads
    constexpr int WGM = NW / 2;             // waves along m (2 along n)
    constexpr int WROWS = TSM / WGM;        // rows per wave
    constexpr int WCOLS = TSN / 2;          // cols per wave
    constexpr int MT = WROWS / 16;          // 16x16 accumulators per wave: MT x NTL
    constexpr int NTL = WCOLS / 16;
    constexpr int TSX = TSM > TSN ? TSM : TSN;
    constexpr int STAGE = (TSX * LD_MK_OF(R) > BK * (TSX + 16)) ? TSX * LD_MK_OF(R) : BK * (TSX + 16);
    constexpr int NCHA = 8 * TSM / NT, NCHB = 8 * TSN / NT;   // 16-byte chunks per thread per stage
    // staged straight into LDS: every k-contiguous operand tile, and 128-wide m-contiguous fp64 ones (a 64-wide
    // m-contiguous k-row, or a float one, is only half a wave-wide load)
    constexpr bool F64 = sizeof(R) == 8;
    constexpr bool ADIR = !A_KM || (F64 && TSM == 128), BDIR = !B_KM || (F64 && TSN == 128);
    static_assert(4 * STAGE == gemm_smem_elems_t<R, TSM, TSN>(), "LDS size");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware bijective remap (block b runs on XCD b%8, in order b/8 on that XCD).
    //  chunk == 0: every XCD gets one contiguous slice of the tile list -- best L2 reuse when all
    //              tiles cost the same (SYRK-shaped trailing updates).
    //  chunk  > 0: the list is dealt to the XCDs in chunks of that many tiles (one 8x8 patch), back
    //              and forth, so lists sorted by decreasing k-range stay balanced across XCDs.
    constexpr int QN = 128 / TSN;                       // workgroups per tile along n
    constexpr int QUADS = (128 / TSM) * QN;             // workgroups per 128x128 tile
    const int n = g.ntiles, b = bx / QUADS, quad = bx % QUADS;
    int p;
    if (QUADS > 1) {
        p = b;
    } else if (g.chunk == 0) {
        const int q = n >> 3, r = n & 7, x = b & 7, yy = b >> 3;
        p = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + yy;
    } else {
        const int C = g.chunk, full = (n / (8 * C)) * (8 * C);
        if (b < full) {
            // serpentine: odd rounds deal in reverse, so that on a list sorted by cost no XCD always
            // gets the most expensive chunk of the round (16 % spread between XCD 0 and 7 otherwise)
            const int x = b & 7, y = b >> 3, round = y / C;
            p = (round * 8 + ((round & 1) ? 7 - x : x)) * C + (y % C);
        } else {
            p = b;
        }
    }
    TileDesc t = g.tiles[p];
    if (g.kfix1 > g.kfix0) { t.kb0 = g.kfix0; t.kb1 = g.kfix1; }
    const int nsteps = (t.kb1 - t.kb0) * (NB / BK);
    const int qi = (quad / QN) * TSM, qj = (quad % QN) * TSN;
    // batch: by selects the problem; operands advance by their per-problem strides.  The pointers of
    // GemmArgs are typed double* on the host side whatever the handle's precision.
    const R* gA = reinterpret_cast<const R*>(g.A) + by * g.sA;
    const R* gB = reinterpret_cast<const R*>(g.B) + by * g.sB;
    R* gC = g.C ? reinterpret_cast<R*>(g.C) + by * g.sC : nullptr;
    if (g.colpart) g.colpart += by * g.sColpart;

    // operand origins (element units)
    const int64_t a_m0 = (int64_t)(t.ci + (A_KM ? g.a_coff : g.a_roff)) * NB + qi;
    const int64_t a_k0 = (int64_t)(t.kb0 + (A_KM ? g.a_roff : g.a_coff)) * NB;
    const int64_t b_n0 = (int64_t)(t.cj + (B_KM ? g.b_coff : g.b_roff)) * NB + qj;
    const int64_t b_k0 = (int64_t)(t.kb0 + (B_KM ? g.b_roff : g.b_coff)) * NB;

    ACC acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = (ACC){0, 0, 0, 0};

    // krev: walk the k-range from its END.  Tiles of one patch whose ranges share their upper end
    // (K^-1 = L^-T L^-1: [ci, nb)) then sweep the same operand rows at the same time, so the panels
    // they share are still in the XCD's L2 when the next tile asks for them.
    const int64_t kfirst = g.krev ? (int64_t)(nsteps - 1) * BK : 0;
    const int64_t kstride = g.krev ? -BK : BK;
    CH ra[NCHA], rb[NCHB];
    if (nsteps > 0) {
        if constexpr (ADIR) stage_direct<R, A_KM, NW, TSM>(gA, g.lda, a_m0, a_k0 + kfirst, smem, wave, lane);
        else stage_load<R, A_KM, NT, TSM>(ra, gA, g.lda, a_m0, a_k0 + kfirst, tid);
        if constexpr (BDIR) stage_direct<R, B_KM, NW, TSN>(gB, g.ldb, b_n0, b_k0 + kfirst, smem + STAGE, wave, lane);
        else stage_load<R, B_KM, NT, TSN>(rb, gB, g.ldb, b_n0, b_k0 + kfirst, tid);
        if constexpr (!ADIR) stage_store<R, A_KM, NT, TSM>(ra, smem, tid);
        if constexpr (!BDIR) stage_store<R, B_KM, NT, TSN>(rb, smem + STAGE, tid);
        if (ADIR || BDIR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        const R* As = smem + (s & 1) * 2 * STAGE;
        const R* Bs = As + STAGE;
        const bool more = (s + 1 < nsteps);
        if (more) {
            // the other stage buffer was last read in step s-1: every wave is past that barrier
            R* An = smem + ((s + 1) & 1) * 2 * STAGE;
            const int64_t koff = kfirst + (int64_t)(s + 1) * kstride;
            if constexpr (ADIR) stage_direct<R, A_KM, NW, TSM>(gA, g.lda, a_m0, a_k0 + koff, An, wave, lane);
            else stage_load<R, A_KM, NT, TSM>(ra, gA, g.lda, a_m0, a_k0 + koff, tid);
            if constexpr (BDIR) stage_direct<R, B_KM, NW, TSN>(gB, g.ldb, b_n0, b_k0 + koff, An + STAGE, wave, lane);
            else stage_load<R, B_KM, NT, TSN>(rb, gB, g.ldb, b_n0, b_k0 + koff, tid);
        }
        // The MFMA block runs at raised wave priority: the arbiter then prefers this wave's MFMAs and
        // fragment reads over the other resident wave's staging instructions, which otherwise steal
        // issue slots from the matrix pipe (tools/gemm_ablate.hip: 66.1 -> 71.2 TFLOP/s for this loop).
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            R a[MT], bb[NTL];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                a[i] = (ADIR && !A_KM) ? frag_mk_swz<R>(As, wm * WROWS + i * 16, kk, lane)
                                       : frag<R, A_KM, TSM>(As, wm * WROWS + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                bb[j] = (BDIR && !B_KM) ? frag_mk_swz<R>(Bs, wn * WCOLS + j * 16, kk, lane)
                                        : frag<R, B_KM, TSN>(Bs, wn * WCOLS + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[i][j] = RT<R>::mfma(a[i], bb[j], acc[i][j]);
        }
        __builtin_amdgcn_s_setprio(0);
        if (more) {
            R* An = smem + ((s + 1) & 1) * 2 * STAGE;
            if constexpr (!ADIR) stage_store<R, A_KM, NT, TSM>(ra, An, tid);
            if constexpr (!BDIR) stage_store<R, B_KM, NT, TSN>(rb, An + STAGE, tid);
            if (ADIR || BDIR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }

    if (EPI == EPI_STORE) {
        // element (16-row band i, reg rg) of the wave's sub-tile sits in row crow0 + i*16 + drow(lane, rg)
        const int64_t crow0 = (int64_t)(t.ci + g.c_roff) * NB + qi + wm * WROWS;
        const int64_t ccol0 = (int64_t)(t.cj + g.c_coff) * NB + qj + wn * WCOLS + (lane & 15);
        const R alpha = (R)g.alpha, beta = (R)g.beta;
        if (g.beta != 0.0) {
            // accumulate into C: fetch one 16-row band of the wave's sub-tile (NTL x 4 values per lane)
            // with all loads in flight, then combine and store -- element by element the loads and
            // stores serialise into 2 x 64 memory round trips per tile
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                R cv[NTL][4];
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        cv[j][rg] = gC[(crow0 + i * 16 + RT<R>::drow(lane, rg)) * g.ldc + ccol0 + j * 16];
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        gC[(crow0 + i * 16 + RT<R>::drow(lane, rg)) * g.ldc + ccol0 + j * 16] = alpha * acc[i][j][rg] + beta * cv[j][rg];
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg)
                        gC[(crow0 + i * 16 + RT<R>::drow(lane, rg)) * g.ldc + ccol0 + j * 16] = alpha * acc[i][j][rg];
        }
    } else {
        // column sums of squares of the 128x128 product tile -> colpart[ci][cj*128 + col]  (full tiles only);
        // accumulated and stored in double for either element type
        double* red = reinterpret_cast<double*>(smem);      // [WGM][128]; all waves are past the last barrier of the k-loop
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) s += (double)acc[i][j][rg] * (double)acc[i][j][rg];
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

template <typename R, bool A_KM, bool B_KM, int EPI, int NW, int TSM, int TSN>
__global__ __launch_bounds__(NW * 64, 2) void gemm_tiles_kernel_t(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) R smem[gemm_smem_elems_t<R, TSM, TSN>()];
    gemm_tile_body_t<R, A_KM, B_KM, EPI, NW, TSM, TSN>(g, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// Launches of a few tiles per CU with very different k-ranges (mid-size N): the launch lasts as long as its
// longest tile, which in the 4-wave shape shares its CU -- and the CU's MFMA rate -- with a second workgroup.
// Up to this many tiles the 8-wave shape is launched with 24 KB of unused dynamic LDS on top of its 74 KB of
// staging buffers, so that every tile has a CU to itself (K^-1 product at N = 4206: 0.71 -> 0.56 ms).
static int mid_tiles() {
    static const int v = 2048;
    return v;
}

template <typename R, bool A_KM, bool B_KM, int EPI>
static int launch_one(gpimhip_ctx* h, const GemmArgs& g) {
    if (g.ntiles <= 0) return GPIMHIP_OK;
    const int64_t total = (int64_t)g.ntiles * h->nbatch;
    const bool small = total <= 256;
    if (EPI == EPI_STORE && small && !g.inplace)
        // few tiles: spread each over four CUs (64x64 quadrants)
        hipLaunchKernelGGL((gemm_tiles_kernel_t<R, A_KM, B_KM, EPI_STORE, 4, 64, 64>), dim3(g.ntiles * 4, h->nbatch),
                           dim3(256), 0, h->stream, g);
    else if (EPI == EPI_STORE && small)
        // in-place panel solve: row halves (the workgroup owns the rows it overwrites), 8 waves
        hipLaunchKernelGGL((gemm_tiles_kernel_t<R, A_KM, B_KM, EPI_STORE, 8, 64, 128>), dim3(g.ntiles * 2, h->nbatch),
                           dim3(512), 0, h->stream, g);
    else if (EPI == EPI_STORE && total > 256 && total <= mid_tiles())
        // one tile per CU (see mid_tiles()).  Not for the column-sum epilogue: its cross-wave summation order
        // follows the wave layout, and batched and stand-alone predictions must stay bit-identical.
        hipLaunchKernelGGL((gemm_tiles_kernel_t<R, A_KM, B_KM, EPI, 8, 128, 128>), dim3(g.ntiles, h->nbatch), dim3(512),
                           24 * 1024, h->stream, g);
    else if (total <= 256 || (!A_KM && !B_KM))
        // (also every SYRK-shaped update of the Cholesky: measured 8 % faster factorisation at N = 16384,
        // the 512-thread workgroups interleave better with the concurrent panel chain)
        // at most one tile per CU: 8-wave workgroup so every SIMD still holds two MFMA waves
        hipLaunchKernelGGL((gemm_tiles_kernel_t<R, A_KM, B_KM, EPI, 8, 128, 128>), dim3(g.ntiles, h->nbatch), dim3(512), 0,
                           h->stream, g);
    else
        hipLaunchKernelGGL((gemm_tiles_kernel_t<R, A_KM, B_KM, EPI, 4, 128, 128>), dim3(g.ntiles, h->nbatch), dim3(256), 0,
                           h->stream, g);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

template <typename R>
static int launch_gemm_t(gpimhip_ctx* h, bool a_km, bool b_km, int epi, const GemmArgs& g) {
    if (epi == EPI_STORE) {
        if (!a_km && !b_km) return launch_one<R, false, false, EPI_STORE>(h, g);   // NT
        if (!a_km && b_km) return launch_one<R, false, true, EPI_STORE>(h, g);     // NN
        if (a_km && b_km) return launch_one<R, true, true, EPI_STORE>(h, g);       // TN
    } else {
        if (!a_km && b_km) return launch_one<R, false, true, EPI_COLSUMSQ>(h, g);
    }
    gpim_set_error("launch_gemm: unsupported operand layout combination");
    return GPIMHIP_E_BADARG;
}

