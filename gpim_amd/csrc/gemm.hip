// gemm.hip -- the fp64 MFMA tile engine of libgpimhip.
//
// One kernel template covers every O(N^3) stage of the exact-GP hot path
// (SURVEY 8(a) rows a6, a8, a11): Cholesky panel solves and trailing SYRK updates,
// the triangular inverse, K^-1 = L^-T L^-1 and the predictive-variance product
// L^-1 K(X, X*).  Work is described by a list of 128x128 output tiles, each with its own
// k-block range, so triangular operands simply get shorter ranges (no wasted MFMAs on
// structural zeros) and the host can order the list for XCD/L2 locality.
//
// Tile engine: 256 threads = 4 waves (2x2), each wave owns a 64x64 sub-tile = 4x4
// v_mfma_f64_16x16x4_f64 accumulators (128 VGPRs).  Two more shapes serve launches that cannot fill
// the chip with one 4-wave workgroup per 128x128 tile: 8 waves per tile (two MFMA waves on every
// SIMD of the tile's CU) and 64x64 sub-tiles (a 128x128 tile spread over 4 CUs) -- the panel
// solves / in-panel updates of the Cholesky are latency-bound chains of such small launches.
// Operand tiles are double-buffered in LDS, one barrier per 16-deep k-step.  128-wide tiles are staged
// global -> LDS directly (global_load_lds_dwordx4, one wave-wide 16-byte load per 1 KB of LDS, no
// VGPR staging and no ds_write); 64-wide tiles go global -> registers -> LDS.  Every global access is
// a coalesced 16-byte load whatever the transpose:
//   "KM" operand (row-major, m contiguous):  lds[16][128+16]; one load per k-row
//   "MK" operand (row-major, k contiguous):  1 KB groups of 8 rows x 8 chunks, chunk index XOR row
//                                            (direct), or lds[64][16+2] (through registers)
// The MFMA block of a k-step runs at raised wave priority (s_setprio).
// v_mfma_f64_16x16x4_f64 lane maps (cdna_hip_programming.md section 3):
//   A[l&15][l>>4], B[l>>4][l&15], C/D col = l&15, row = (l>>4) + 4*reg.
#include <stdlib.h>
#include "gemm_body.hpp"

template <bool A_KM, bool B_KM, int EPI, int NW, int TSM, int TSN>
__global__ __launch_bounds__(NW * 64, 2) void gemm_tiles_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double smem[gemm_smem_doubles<TSM, TSN>()];
    gemm_tile_body<A_KM, B_KM, EPI, NW, TSM, TSN>(g, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// Launches of a few tiles per CU with very different k-ranges (mid-size N): the launch lasts as long as its
// longest tile, which in the 4-wave shape shares its CU -- and the CU's MFMA rate -- with a second workgroup.
// Up to this many tiles the 8-wave shape is launched with 24 KB of unused dynamic LDS on top of its 74 KB of
// staging buffers, so that every tile has a CU to itself (K^-1 product at N = 4206: 0.71 -> 0.56 ms).
static int mid_tiles() {
    static const int v = 1100;   // (2048-tile levels of the inverse at N = 16384: 4-wave, two per CU, is 0.25 ms faster)
    return v;
}

template <bool A_KM, bool B_KM, int EPI>
static int launch_one(gpimhip_ctx* h, const GemmArgs& g) {
    if (g.ntiles <= 0) return GPIMHIP_OK;
    const int64_t total = (int64_t)g.ntiles * h->nbatch / (g.shape_div > 1 ? g.shape_div : 1);
    // up to 640 tiles: 64x64 quadrants, four co-resident workgroups per CU.  A launch of a few hundred tiles whose
    // k-ranges differ by an order of magnitude (triangular inverse, K^-1 product at N ~ 4000) lasts as long as its
    // longest tile; dealt longest-first over 4 x 256 slots, every CU gets a mix (N = 4206: inverse 0.92 -> 0.80 ms,
    // K^-1 product 0.57 -> 0.48; 1200 / 2400 measure the same; round 6: 128x64 halves with 8 waves instead of quadrants
    // 2.303 against 2.292 ms per Adam iteration at N = 4212, 1.144 against 1.084 at N = 2560, the same at 6000)
    const int tile64_max = 640;
    const bool small = total <= tile64_max;
    if (EPI == EPI_STORE && small && !g.inplace)
        // few tiles: spread each over four CUs (64x64 quadrants)
        hipLaunchKernelGGL((gemm_tiles_kernel<A_KM, B_KM, EPI_STORE, 4, 64, 64>), dim3(g.ntiles * 4, h->nbatch),
                           dim3(256), 0, h->stream, g);
    else if (EPI == EPI_STORE && small)
        // in-place panel solve: row halves (the workgroup owns the rows it overwrites), 8 waves
        hipLaunchKernelGGL((gemm_tiles_kernel<A_KM, B_KM, EPI_STORE, 8, 64, 128>), dim3(g.ntiles * 2, h->nbatch),
                           dim3(512), 0, h->stream, g);
    else if (EPI == EPI_STORE && total > 256 && total <= mid_tiles())
        // one tile per CU (see mid_tiles()).  Not for the column-sum epilogue: its cross-wave summation order
        // follows the wave layout, and batched and stand-alone predictions must stay bit-identical.
        hipLaunchKernelGGL((gemm_tiles_kernel<A_KM, B_KM, EPI, 8, 128, 128>), dim3(g.ntiles, h->nbatch), dim3(512),
                           24 * 1024, h->stream, g);
    else if (total <= 256 || (!A_KM && !B_KM))
        // (also every SYRK-shaped update of the Cholesky: measured 8 % faster factorisation at N = 16384,
        // the 512-thread workgroups interleave better with the concurrent panel chain)
        // at most one tile per CU: 8-wave workgroup so every SIMD still holds two MFMA waves
        hipLaunchKernelGGL((gemm_tiles_kernel<A_KM, B_KM, EPI, 8, 128, 128>), dim3(g.ntiles, h->nbatch), dim3(512), 0,
                           h->stream, g);
    else
        hipLaunchKernelGGL((gemm_tiles_kernel<A_KM, B_KM, EPI, 4, 128, 128>), dim3(g.ntiles, h->nbatch), dim3(256), 0,
                           h->stream, g);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// Single-precision handles (gpimhip_set_precision) run the float instantiation of the same engine, written once
// for both element types in gemm_kernel.hpp and compiled in gemm32.hip.  The double kernels stay the hand-tuned
// source of this file: instantiating the generic template for double gives the same instruction counts but a
// schedule that is 0.5 - 0.9 % slower on the triangular inverse and the factorisation at N = 16384 (A/B on one box).
int launch_gemm_f32(gpimhip_ctx* h, bool a_km, bool b_km, int epi, const GemmArgs& g);
static int launch_gemm_f64(gpimhip_ctx* h, bool a_km, bool b_km, int epi, const GemmArgs& g) {
    if (epi == EPI_STORE) {
        if (!a_km && !b_km) return launch_one<false, false, EPI_STORE>(h, g);   // NT
        if (!a_km && b_km) return launch_one<false, true, EPI_STORE>(h, g);     // NN
        if (a_km && b_km) return launch_one<true, true, EPI_STORE>(h, g);       // TN
    } else {
        if (!a_km && b_km) return launch_one<false, true, EPI_COLSUMSQ>(h, g);
    }
    gpim_set_error("launch_gemm: unsupported operand layout combination");
    return GPIMHIP_E_BADARG;
}

int launch_gemm(gpimhip_ctx* h, bool a_km, bool b_km, int epi, const GemmArgs& g) {
    return h->fp32 ? launch_gemm_f32(h, a_km, b_km, epi, g) : launch_gemm_f64(h, a_km, b_km, epi, g);
}
